"""GPU parity of the resident key sets (fa_keyset_create / fa_verify_many_keyset): the decoded-key form of
verification::verify<N, beta2>(h, ...) (verification.hpp:20-26, falcon.hpp:226-229, caller bench_signing.cpp:105) for
host callers that keep their keys across calls. Everything is held to the CPU oracle on the same inputs."""
import threading

import numpy as np
import pytest

import support
from test_gpu_parity import _compare_debug

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("logn", [9, 10])
def test_keyset_against_oracle(ctx, oracle, corpus, logn):
    """Keys resident from their encoded form: one key per item (key_idx = None) including a key with a wrong header
    and keys with coefficients >= q (malformed classes 2 and 12), then a few keys reused by many items through an
    index, in calls of every size class (one wave per item, small fronts, sub-batches of the staging pipeline)."""
    from falcon_amd import capi
    count, nkeys = 5000, 37
    pk, msg, sig, kinds = corpus.generate(logn, count, seed=61 + logn, malformed_permille=300)
    assert (kinds == 2).any() and (kinds == 12).any()  # bad_pk_header, pk_coeff_ge_q
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    ks = ctx.keyset_create(logn, pk=pk)
    try:
        assert len(ks) == count and ks.logn == logn and ks.device_ptr
        acc, gdbg = ctx.verify_many_keyset(ks, msg, sig, debug=True)
        _compare_debug(logn, flags, odbg, acc, gdbg)
        np.testing.assert_array_equal(ctx.verify_many_keyset(ks, msg, sig, norm_policy=1), (flags >> 1) & 1)
        # key_idx reuse: item i against key i % nkeys; most fail the norm, the first nkeys keep their own key
        kidx = (np.arange(count) % nkeys).astype(np.uint32)
        want, wdbg = oracle.verify_many(logn, pk[kidx], msg, sig, debug=True, threads=8)
        acc2, gdbg2 = ctx.verify_many_keyset(ks, msg, sig, key_idx=kidx, debug=True)
        _compare_debug(logn, want, wdbg, acc2, gdbg2)
        assert acc2[:nkeys].tolist() == (flags[:nkeys] & 1).tolist()
        for m in (1, 7, 200, 1500):   # the zero-copy tiny path, the small fronts
            np.testing.assert_array_equal(ctx.verify_many_keyset(ks, msg[:m], sig[:m]), flags[:m] & 1)
            np.testing.assert_array_equal(ctx.verify_many_keyset(ks, msg[:m], sig[:m], key_idx=kidx[:m]), want[:m] & 1)
        # ragged messages through offsets
        lens = (np.arange(300) * 7) % 150
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        flat = np.random.default_rng(4).integers(0, 256, int(off[-1]), dtype=np.uint8)
        w3, _ = oracle.verify_many(logn, pk[:300], flat, sig[:300], msg_off=off, threads=8)
        np.testing.assert_array_equal(ctx.verify_many_keyset(ks, flat, sig[:300], msg_off=off), w3 & 1)
        # a private context with small sub-batches: the two-set staging pipeline with resident keys, pinned inputs too
        c = capi.Context(0)
        try:
            c.set_option(capi.OPT_HOST_CHUNK_ITEMS, 512)
            c.set_option(capi.OPT_TINY_BATCH_ITEMS, 0)
            np.testing.assert_array_equal(c.verify_many_keyset(ks, msg, sig), flags & 1)   # another context of the same GPU
            np.testing.assert_array_equal(c.verify_many_keyset(ks, msg, sig, key_idx=kidx), want & 1)
            pin = [capi.PinnedArray(a.shape) for a in (msg, sig)] + [capi.PinnedArray((count,))]
            pin[0].array[...] = msg
            pin[1].array[...] = sig
            c.verify_many_keyset(ks, pin[0].array, pin[1].array, key_idx=kidx, accept_out=pin[2].array)
            np.testing.assert_array_equal(pin[2].array, want & 1)
            for b in pin:
                b.close()
        finally:
            c.close()
        # refused: an index past the set, more items than keys without an index
        with pytest.raises(capi.FalconAmdError):
            ctx.verify_many_keyset(ks, msg, sig, key_idx=kidx + np.uint32(count))
        ks_small = ctx.keyset_create(logn, pk=pk[:nkeys])
        try:
            with pytest.raises(capi.FalconAmdError):
                ctx.verify_many_keyset(ks_small, msg, sig)
            np.testing.assert_array_equal(ctx.verify_many_keyset(ks_small, msg, sig, key_idx=kidx), want & 1)
        finally:
            ks_small.close()
    finally:
        ks.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_keyset_from_decoded_polynomials_and_device_calls(ctx, oracle, corpus, logn):
    """The set made from decoded polynomials h (what verification::verify takes) gives the same answers wherever the
    key's header was right; its device pointer serves the device-resident prepared-key call."""
    import torch
    count = 700
    pk, msg, sig, _ = corpus.generate(logn, count, seed=71 + logn, malformed_permille=200, kind=-1)
    good = pk[:, 0] == logn
    h = np.stack([oracle.decode_pkey(logn, k)[1] for k in pk[good]])
    flags, _ = oracle.verify_many(logn, pk[good], msg[good], sig[good], threads=8)
    ks = ctx.keyset_create(logn, h=h)
    try:
        np.testing.assert_array_equal(ctx.verify_many_keyset(ks, msg[good], sig[good]), flags & 1)
        dev = torch.device("cuda", 0)
        d_msg, d_sig = (torch.from_numpy(np.ascontiguousarray(a[good])).to(dev) for a in (msg, sig))
        d_acc = torch.zeros(int(good.sum()), dtype=torch.uint8, device=dev)
        ctx.verify_many_prepared_dev(logn, int(good.sum()), ks.device_ptr, None, d_msg.data_ptr(), None, msg.shape[1],
                                     d_sig.data_ptr(), d_acc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        np.testing.assert_array_equal(d_acc.cpu().numpy(), flags & 1)
    finally:
        ks.close()


def test_keyset_shared_by_threads(oracle, corpus):
    """One set, four host threads, each with a context of its own: the set is read-only."""
    from falcon_amd import capi
    pk, msg, sig, _ = corpus.generate(9, 4000, seed=91, malformed_permille=250)
    flags, _ = oracle.verify_many(9, pk, msg, sig, threads=8)
    c0 = capi.Context(0)
    ks = c0.keyset_create(9, pk=pk)
    errs = []

    def work(t):
        c = capi.Context(0)
        try:
            lo, hi = t * 1000, (t + 1) * 1000
            idx = np.arange(lo, hi, dtype=np.uint32)
            for _ in range(5):
                if not np.array_equal(c.verify_many_keyset(ks, msg[lo:hi], sig[lo:hi], key_idx=idx), flags[lo:hi] & 1):
                    errs.append(t)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
        finally:
            c.close()
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ks.close()
    c0.close()
    assert not errs, errs
