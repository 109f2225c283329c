"""GPU: the device-resident mixed-N entry point (fa_verify_mixed_dev), asynchronous calls with the device-side
retry pass (FA_OPT_ASYNC, fa_finish), and calls on two streams through one context."""
import numpy as np
import pytest

import mixed_batch
import support

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch, torch.device("cuda", 0)


def _mixed_inputs(corpus, oracle, npairs, seed, permille=300):
    p9 = corpus.generate(9, npairs, seed=seed, malformed_permille=permille, threads=16)
    p10 = corpus.generate(10, npairs, seed=seed + 1, malformed_permille=permille, threads=16)
    bits = mixed_batch.pair_order_bits(npairs, seed)
    logn, pk_off, sig_off, msg_off, src = mixed_batch.layout(bits, 32)
    pk = mixed_batch.pack_numpy(bits, p9[0], p10[0])
    sig = mixed_batch.pack_numpy(bits, p9[2], p10[2])
    msg = mixed_batch.pack_numpy(bits, p9[1], p10[1])
    return p9, p10, logn, pk, pk_off, msg, msg_off, sig, sig_off, src


def _want(oracle, p9, p10, logn, src, nsample=None):
    w9 = oracle.verify_many(9, *p9[:3], threads=16)[0]
    w10 = oracle.verify_many(10, *p10[:3], threads=16)[0]
    return np.where(logn == 9, w9[src], w10[src])


@pytest.mark.parametrize("npairs", [1, 37, 3000])
def test_mixed_dev_against_oracle(ctx, oracle, corpus, npairs):
    torch, dev = _torch()
    p9, p10, logn, pk, pk_off, msg, msg_off, sig, sig_off, src = _mixed_inputs(corpus, oracle, npairs, 900 + npairs)
    want = _want(oracle, p9, p10, logn, src)
    t = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev)
         for k, v in dict(logn=logn, pk=pk, pk_off=pk_off, msg=msg, msg_off=msg_off, sig=sig, sig_off=sig_off).items()}
    for policy in (0, 1):
        acc = torch.full((2 * npairs,), 7, dtype=torch.uint8, device=dev)
        ctx.verify_mixed_dev(2 * npairs, t["logn"].data_ptr(), t["pk"].data_ptr(), t["pk_off"].data_ptr(), t["msg"].data_ptr(),
                             t["msg_off"].data_ptr(), t["sig"].data_ptr(), t["sig_off"].data_ptr(), acc.data_ptr(),
                             norm_policy=policy, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(acc.cpu().numpy(), (want >> policy) & 1)
    # the host-buffer entry point runs the same path
    acc_h = ctx.verify_mixed(logn, pk, pk_off, msg, msg_off, sig, sig_off)
    np.testing.assert_array_equal(acc_h, want & 1)


def test_mixed_dev_unaligned_records_and_single_class(ctx, oracle, corpus):
    """Records at odd byte offsets with gaps between them; batches that hold one class only; a wrong tag."""
    torch, dev = _torch()
    from falcon_amd import capi
    rng = np.random.default_rng(5)
    for classes in ((9, 10), (9,), (10,)):
        items = []
        for ln in classes:
            pk, msg, sig, _ = corpus.generate(ln, 90, seed=40 + ln, malformed_permille=300, mlen=21)
            w = oracle.verify_many(ln, pk, msg, sig)[0] & 1
            items += [(ln, pk[i], msg[i], sig[i], w[i]) for i in range(90)]
        order = rng.permutation(len(items))
        logn, want = np.array([items[o][0] for o in order], np.uint8), np.array([items[o][4] for o in order], np.uint8)
        def scatter(col):
            buf, offs, pos = [], [], 0
            for o in order:
                gap = int(rng.integers(0, 9))
                buf.append(rng.integers(0, 256, gap, dtype=np.uint8))
                pos += gap
                offs.append(pos)
                buf.append(items[o][col])
                pos += items[o][col].size
            return np.concatenate(buf + [np.zeros(16, np.uint8)]), np.array(offs, np.uint64)
        pk, pk_off = scatter(1)
        sig, sig_off = scatter(3)
        msg = np.concatenate([items[o][2] for o in order])
        msg_off = np.arange(len(items) + 1, dtype=np.uint64) * np.uint64(21)
        acc = ctx.verify_mixed(logn, pk, pk_off, msg, msg_off, sig, sig_off)
        np.testing.assert_array_equal(acc, want)
    bad = logn.copy()
    bad[3] = 11
    with pytest.raises(capi.FalconAmdError) as e:
        ctx.verify_mixed(bad, pk, pk_off, msg, msg_off, sig, sig_off)
    assert e.value.code == capi.E_LOGN


@pytest.mark.parametrize("logn,blocks", [(9, 8), (10, 16)])
def test_async_calls_and_device_side_retry(oracle, corpus, logn, blocks):
    """FA_OPT_ASYNC: calls return without synchronising; with too few SHAKE256 blocks up front (8 / 16: about half of
    the items need one more) the retry pass enqueued behind the batch settles them on the device; a batch with more
    such items than the pass takes reports them through fa_finish and leaves their accept byte 0."""
    torch, dev = _torch()
    from falcon_amd import capi
    c = capi.Context(0)
    try:
        opt = capi.OPT_HASH_BLOCKS_512 if logn == 9 else capi.OPT_HASH_BLOCKS_1024
        stream = torch.cuda.current_stream().cuda_stream
        big = corpus.generate(logn, 12000, seed=77, malformed_permille=200, threads=16)
        small = corpus.generate(logn, 3000, seed=78, malformed_permille=200, threads=16)
        want_big = oracle.verify_many(logn, *big[:3], threads=16)[0] & 1
        want_small = oracle.verify_many(logn, *small[:3], threads=16)[0] & 1
        tb = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in big[:3]]
        ts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in small[:3]]
        acc_b = torch.full((12000,), 7, dtype=torch.uint8, device=dev)
        acc_s = torch.full((3000,), 7, dtype=torch.uint8, device=dev)
        c.set_option(capi.OPT_ASYNC, 1)
        # default block counts: nothing to retry; two calls queued back to back, one finish
        c.verify_many_dev(logn, 12000, tb[0].data_ptr(), tb[1].data_ptr(), None, 32, tb[2].data_ptr(), acc_b.data_ptr(), stream=stream)
        c.verify_many_dev(logn, 3000, ts[0].data_ptr(), ts[1].data_ptr(), None, 32, ts[2].data_ptr(), acc_s.data_ptr(), stream=stream)
        assert c.finish(stream) == 0
        np.testing.assert_array_equal(acc_b.cpu().numpy(), want_big)
        np.testing.assert_array_equal(acc_s.cpu().numpy(), want_small)
        # too few blocks: the small batch's queued items (about half of 3000) fit the retry pass of a workspace that
        # has held 12000 items (capacity min(4096, 12000 / 2))
        c.set_option(opt, blocks)
        acc_s.fill_(7)
        c.verify_many_dev(logn, 3000, ts[0].data_ptr(), ts[1].data_ptr(), None, 32, ts[2].data_ptr(), acc_s.data_ptr(), stream=stream)
        assert c.finish(stream) == 0
        np.testing.assert_array_equal(acc_s.cpu().numpy(), want_small)
        # the large batch queues more than 4096 items: the excess is reported, never silently accepted
        acc_b.fill_(7)
        c.verify_many_dev(logn, 12000, tb[0].data_ptr(), tb[1].data_ptr(), None, 32, tb[2].data_ptr(), acc_b.data_ptr(), stream=stream)
        unresolved = c.finish(stream)
        got = acc_b.cpu().numpy()
        assert unresolved > 0
        assert ((got == want_big) | (got == 0)).all() and int((got != want_big).sum()) <= unresolved
        assert c.finish(stream) == 0  # the count is reset by fa_finish
        # the synchronous call settles everything whatever the block count
        c.set_option(capi.OPT_ASYNC, 0)
        acc_b.fill_(7)
        c.verify_many_dev(logn, 12000, tb[0].data_ptr(), tb[1].data_ptr(), None, 32, tb[2].data_ptr(), acc_b.data_ptr(), stream=stream)
        np.testing.assert_array_equal(acc_b.cpu().numpy(), want_big)
    finally:
        c.close()


def test_two_streams_share_one_context(ctx, oracle, corpus):
    """Calls enqueued on different streams through one context use one workspace: the library orders them on the
    device (fa_decode_sigs_dev on stream A followed by fa_verify_many_dev on stream B)."""
    torch, dev = _torch()
    from falcon_amd import capi
    lib = capi.load_library()
    n = 20000
    pk, msg, sig, _ = corpus.generate(9, n, seed=31, malformed_permille=200, threads=16)
    pk2, msg2, sig2, _ = corpus.generate(9, n, seed=32, malformed_permille=200, threads=16)
    want2 = oracle.verify_many(9, pk2, msg2, sig2, threads=16)[0] & 1
    salt_ref, s2_ref, ok_ref = ctx.decode_sigs(9, sig)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    d_sig = torch.from_numpy(sig).to(dev)
    d2 = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk2, msg2, sig2)]
    torch.cuda.synchronize()
    for _ in range(3):
        d_salt = torch.zeros((n, 40), dtype=torch.uint8, device=dev)
        d_s2 = torch.zeros((n, 512), dtype=torch.int16, device=dev)
        d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        acc = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        rc = lib.fa_decode_sigs_dev(ctx.handle, 9, n, d_sig.data_ptr(), d_salt.data_ptr(), d_s2.data_ptr(), d_ok.data_ptr(), sa.cuda_stream)
        assert rc == 0
        ctx.verify_many_dev(9, n, d2[0].data_ptr(), d2[1].data_ptr(), None, 32, d2[2].data_ptr(), acc.data_ptr(), stream=sb.cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_ok.cpu().numpy(), ok_ref)
        np.testing.assert_array_equal(d_s2.cpu().numpy(), s2_ref)
        np.testing.assert_array_equal(d_salt.cpu().numpy(), salt_ref)
        np.testing.assert_array_equal(acc.cpu().numpy(), want2)


def test_failed_call_leaves_the_workspace_fence_whole(oracle, corpus):
    """A call that fails after it has begun using the context's workspace -- a tag array with a wrong tag, found only
    after the partition kernels ran; a key buffer of an absurd size in the specification wire format -- must still leave
    the fence behind what it enqueued: good calls on ANOTHER stream of the same context follow, and match the oracle."""
    torch, dev = _torch()
    from falcon_amd import capi
    c = capi.Context(0)
    try:
        pk, msg, sig, _ = corpus.generate(9, 3000, seed=123, malformed_permille=300, threads=16)
        want = oracle.verify_many(9, pk, msg, sig, threads=16)[0] & 1
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)]
        acc = torch.zeros(3000, dtype=torch.uint8, device=dev)
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        # (1) mixed batch with a tag that is neither 9 nor 10
        n = 5000
        tags = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        tags[n // 2] = 11
        off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        with pytest.raises(capi.FalconAmdError) as e:
            c.verify_mixed_dev(n, tags.data_ptr(), t[0].data_ptr(), off.data_ptr(), t[1].data_ptr(), off.data_ptr(), t[2].data_ptr(),
                               off.data_ptr(), acc.data_ptr(), stream=s1.cuda_stream)
        assert e.value.code == capi.E_LOGN
        c.verify_many_dev(9, 3000, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=s2.cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(acc.cpu().numpy(), want)
        # (2) specification wire format: the repacked keys of 2^32 - 1 items do not fit any GPU; the call fails in its reserve
        c.set_option(capi.OPT_WIRE_FORMAT, capi.WIRE_SPEC)
        with pytest.raises(capi.FalconAmdError):
            c.verify_many_dev(9, 0xFFFFFFFF, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=s1.cuda_stream)
        c.set_option(capi.OPT_WIRE_FORMAT, capi.WIRE_REFERENCE)
        acc.fill_(7)
        torch.cuda.synchronize()
        c.verify_many_dev(9, 3000, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=s2.cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(acc.cpu().numpy(), want)
    finally:
        c.close()
