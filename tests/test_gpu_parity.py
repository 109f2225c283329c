"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs."""
import json
import os

import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _compare_debug(logn, flags, odbg, acc, gdbg, policy_bit=0):
    n = 1 << logn
    np.testing.assert_array_equal(acc, (flags >> policy_bit) & 1)
    np.testing.assert_array_equal(gdbg["status"] if policy_bit == 0 else gdbg["status"] * 0, odbg["status"] if policy_bit == 0 else odbg["status"] * 0)
    ok = odbg["status"] != support.ST_BAD_PK_HEADER
    ok &= odbg["status"] != support.ST_BAD_SIG_HEADER
    ok &= odbg["status"] != support.ST_BAD_ENCODING
    np.testing.assert_array_equal(gdbg["norm_exact"], odbg["norm_exact"])
    np.testing.assert_array_equal(gdbg["norm_wrap32"], odbg["norm_wrap32"])
    np.testing.assert_array_equal(gdbg["c"][ok], odbg["c"][ok][:, :n])
    np.testing.assert_array_equal(gdbg["s2"][ok], odbg["s2"][ok][:, :n])
    np.testing.assert_array_equal(gdbg["s1"][ok], odbg["s1"][ok][:, :n])
    assert not gdbg["c"][~ok].any() and not gdbg["s1"][~ok].any() and not gdbg["s2"][~ok].any()


def test_readme_known_answer(ctx, oracle):
    """The only known-answer triple the reference holds (README.md:308-312): must verify true."""
    k = json.load(open(os.path.join(GOLDEN, "readme_kat_falcon512.json")))
    pk, msg, sig = (np.frombuffer(bytes.fromhex(k[x]), np.uint8)[None, :] for x in ("pkey", "msg", "sig"))
    acc, dbg = ctx.verify_many(9, pk, msg, sig, debug=True)
    assert acc[0] == 1 and dbg["status"][0] == 0
    rec = k["survey_recorded"]
    assert int(dbg["norm_exact"][0]) == rec["norm_exact"]
    assert dbg["c"][0, :4].tolist() == rec["c_first4"] and dbg["s2"][0, :4].tolist() == rec["s2_first4"]
    import falcon_amd
    assert falcon_amd.verify(512, pk[0].tobytes(), msg[0].tobytes(), sig[0].tobytes()) is True
    bad = msg.copy()
    bad[0, 0] ^= 1
    assert falcon_amd.verify(512, pk[0].tobytes(), bad[0].tobytes(), sig[0].tobytes()) is False


@pytest.mark.parametrize("logn", [9, 10])
def test_valid_corpus(ctx, oracle, corpus, logn):
    pk, msg, sig, _ = corpus.generate(logn, 3000, seed=101 + logn)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    acc, gdbg = ctx.verify_many(logn, pk, msg, sig, debug=True)
    assert acc.all()
    _compare_debug(logn, flags, odbg, acc, gdbg)
    acc2 = ctx.verify_many(logn, pk, msg, sig)  # non-debug kernel
    np.testing.assert_array_equal(acc2, acc)


@pytest.mark.parametrize("logn", [9, 10])
@pytest.mark.parametrize("kind", range(1, 13))
def test_malformed_classes(ctx, oracle, corpus, logn, kind):
    pk, msg, sig, _ = corpus.generate(logn, 257, seed=500 + kind, kind=kind)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    acc, gdbg = ctx.verify_many(logn, pk, msg, sig, debug=True)
    _compare_debug(logn, flags, odbg, acc, gdbg)
    accw = ctx.verify_many(logn, pk, msg, sig, norm_policy=1)  # wrap32 == the g++-built reference
    np.testing.assert_array_equal(accw, (flags >> 1) & 1)


# ---- edge cases: ragged and long messages, empty and tiny batches, chunking, the re-hash pass ----

@pytest.mark.parametrize("logn", [9, 10])
def test_message_lengths_and_ragged_offsets(ctx, oracle, corpus, logn):
    """mlen 0..1024 covers single- and multi-block absorbs (tests/test_signing.cpp:18-19 signs mlen 1..1024);
    the last batch mixes all lengths through the offset-indexed form."""
    lens = [0, 1, 32, 95, 96, 135, 136, 137, 271, 1024]
    pks, sigs, msgs = [], [], []
    for j, mlen in enumerate(lens):
        pk, msg, sig, _ = corpus.generate(logn, 33, seed=40 + j, mlen=mlen, malformed_permille=200)
        flags, _ = oracle.verify_many(logn, pk, msg, sig, threads=8)
        acc = ctx.verify_many(logn, pk, msg, sig)
        np.testing.assert_array_equal(acc, flags & 1, err_msg=f"mlen={mlen}")
        pks.append(pk); sigs.append(sig); msgs.extend(list(msg))
    pk, sig = np.concatenate(pks), np.concatenate(sigs)
    off = np.concatenate([[0], np.cumsum([m.size for m in msgs])]).astype(np.uint64)
    flat = np.concatenate([m for m in msgs if m.size] or [np.zeros(0, np.uint8)])
    flags, odbg = oracle.verify_many(logn, pk, flat, sig, msg_off=off, debug=True, threads=8)
    acc, gdbg = ctx.verify_many(logn, pk, flat, sig, msg_off=off, debug=True)
    _compare_debug(logn, flags, odbg, acc, gdbg)
    from falcon_amd import capi
    c = capi.Context(0)  # the default context takes this size one wave per item; the same through the other small-pass fronts
    try:
        for pair in (0, 1 << 20):
            c.set_option(capi.OPT_TINY_BATCH_ITEMS, 0)
            c.set_option(capi.OPT_PAIR_HASH_ITEMS, pair)
            acc, gdbg = c.verify_many(logn, pk, flat, sig, msg_off=off, debug=True)
            _compare_debug(logn, flags, odbg, acc, gdbg)
    finally:
        c.close()


def test_empty_and_tiny_batches(ctx, oracle, corpus):
    pk, msg, sig, _ = corpus.generate(9, 3, seed=9)
    assert ctx.verify_many(9, pk[:0], msg[:0], sig[:0]).shape == (0,)
    for n in (1, 2, 3):
        np.testing.assert_array_equal(ctx.verify_many(9, pk[:n], msg[:n], sig[:n]), np.ones(n, np.uint8))


@pytest.mark.parametrize("logn", [9, 10])
def test_chunked_passes_and_rehash(oracle, corpus, logn):
    """A private context with 64-item passes and too few squeeze blocks up front: every pass boundary and the
    re-hash pass (items whose rejection sampling runs out of blocks) are exercised."""
    from falcon_amd import capi
    c = capi.Context(0)
    try:
        c.set_option(capi.OPT_CHUNK_ITEMS, 64)
        c.set_option(capi.OPT_HASH_BLOCKS_512 if logn == 9 else capi.OPT_HASH_BLOCKS_1024, 8 if logn == 9 else 16)
        pk, msg, sig, _ = corpus.generate(logn, 1000, seed=77, malformed_permille=300)
        flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
        acc, gdbg = c.verify_many(logn, pk, msg, sig, debug=True)
        _compare_debug(logn, flags, odbg, acc, gdbg)
        prof = c.get_profile()
        need_more = int((odbg["hash_blocks"] > (8 if logn == 9 else 16)).sum())
        assert need_more > 0 and prof["retried_items"] >= need_more
    finally:
        c.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_front_end_paths_and_grids(oracle, corpus, logn):
    """The same batch through the front ends -- decoder and hash waves in one kernel (small passes; one sponge per lane
    or per lane pair) and one kernel after the other (full passes) -- and with odd grids of the one-wave-per-item kernels (one workgroup, a
    prime per CU, more workgroups than items need): identical to the oracle each time."""
    from falcon_amd import capi
    pk, msg, sig, _ = corpus.generate(logn, 777, seed=31 + logn, malformed_permille=300)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    c = capi.Context(0)
    try:
        for small, grid, pair, tiny in ((0, 0, 0, 0), (1 << 20, 0, 0, 0), (1 << 20, 0, 1 << 20, 0), (1 << 20, 0, 0, 1 << 20), (0, 1, 0, 0),
                                        (1 << 20, 7, 1 << 20, 0), (0, 65536, 0, 0)):
            c.set_option(capi.OPT_SMALL_BATCH_ITEMS, small)
            c.set_option(capi.OPT_GRID_PER_CU, grid)
            c.set_option(capi.OPT_PAIR_HASH_ITEMS, pair)  # small passes: one sponge per lane or per lane pair ...
            c.set_option(capi.OPT_TINY_BATCH_ITEMS, tiny)  # ... or one wave per item for decoder and sponge
            acc, gdbg = c.verify_many(logn, pk, msg, sig, debug=True)
            _compare_debug(logn, flags, odbg, acc, gdbg)
            np.testing.assert_array_equal(c.verify_many(logn, pk, msg, sig), acc)
    finally:
        c.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_tiny_batches_every_count(ctx, oracle, corpus, logn):
    """Counts 1..65 and 1024 / 512 (the last size of the one-wave-per-item front) with every fourth item malformed:
    decoded signature, hash-to-point, s1, norm and status identical to the oracle (falcon.hpp:230-249 one item at a
    time is the case this path exists for)."""
    last = 1024 if logn == 9 else 512
    pk, msg, sig, _ = corpus.generate(logn, last, seed=900 + logn, malformed_permille=250)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    for n in list(range(1, 66)) + [last]:
        acc, gdbg = ctx.verify_many(logn, pk[:n], msg[:n], sig[:n], debug=True)
        _compare_debug(logn, flags[:n], odbg[:n], acc, gdbg)


@pytest.mark.parametrize("kind", range(1, 13))
@pytest.mark.parametrize("logn", [9, 10])
def test_tiny_batches_malformed_classes(ctx, oracle, corpus, logn, kind):
    """Every malformed class of the corpus (k >= 8 windows, -0, trailing bits, truncated codes ...) through the
    wave-parallel decoder, which hands the shapes it does not take to the generic one inside the same wave."""
    pk, msg, sig, _ = corpus.generate(logn, 64, seed=700 + kind, kind=kind)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    acc, gdbg = ctx.verify_many(logn, pk, msg, sig, debug=True)
    _compare_debug(logn, flags, odbg, acc, gdbg)


@pytest.mark.parametrize("per_class", [150, 9000])
def test_mixed_batch(ctx, oracle, corpus, per_class):
    """fa_verify_mixed: Falcon-512 and Falcon-1024 items interleaved, split by N on the host; the larger
    case is gathered by several host threads."""
    rng = np.random.default_rng(3)
    n = per_class
    p9 = corpus.generate(9, n, seed=1, malformed_permille=300)
    p10 = corpus.generate(10, n, seed=2, malformed_permille=300)
    want9 = oracle.verify_many(9, *p9[:3], threads=8)[0] & 1
    want10 = oracle.verify_many(10, *p10[:3], threads=8)[0] & 1
    order = rng.permutation(2 * n)
    logn, pk, sig, msg, want = [], [], [], [], []
    for o in order:
        src, w, ln, i = (p9, want9, 9, o) if o < n else (p10, want10, 10, o - n)
        logn.append(ln); pk.append(src[0][i]); msg.append(src[1][i]); sig.append(src[2][i]); want.append(w[i])
    cat = lambda xs: (np.concatenate(xs), np.concatenate([[0], np.cumsum([x.size for x in xs])]).astype(np.uint64))
    (pkb, pko), (msgb, msgo), (sigb, sigo) = cat(pk), cat(msg), cat(sig)
    acc = ctx.verify_mixed(np.array(logn, np.uint8), pkb, pko[:-1], msgb, msgo, sigb, sigo[:-1])
    np.testing.assert_array_equal(acc, np.array(want, np.uint8))


def test_argument_errors(ctx):
    from falcon_amd import capi
    lib = capi.load_library()
    z = np.zeros(4096, np.uint8)
    acc = np.zeros(4, np.uint8)
    p = lambda a: a.ctypes.data
    assert lib.fa_verify_many(ctx.handle, 8, 1, p(z), p(z), None, 32, p(z), 0, p(acc), None) == -2   # FA_E_LOGN
    assert lib.fa_verify_many(ctx.handle, 9, 1, None, p(z), None, 32, p(z), 0, p(acc), None) == -1   # FA_E_ARG
    assert lib.fa_verify_many(ctx.handle, 9, 1, p(z), p(z), None, 32, p(z), 7, p(acc), None) == -1   # bad policy
    assert lib.fa_verify_many_dev(ctx.handle, 9, 1, 0x1001, 0x1000, None, 32, 0x1000, 0, 0x1000, None, None) == -3  # FA_E_ALIGN
    assert lib.fa_set_option(ctx.handle, 99, 1) == -1


# ---- decoded-key fast path and standalone hash_to_point ----

@pytest.mark.parametrize("logn", [9, 10])
def test_prepared_keys_many_signatures_per_key(ctx, oracle, corpus, logn):
    """verification::verify<N, beta2>(h, ...) (verification.hpp:20-26): keys prepared once from their encoded
    form and from the decoded polynomial, items name their key by index; results must equal the packed-key path
    and the oracle, including a key with a wrong header and keys with coefficients >= q."""
    n = 1 << logn
    count, nkeys = 600, 40
    pk, msg, sig, kinds = corpus.generate(logn, count, seed=31 + logn, malformed_permille=300)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    # one prepared key per item (key_idx = None)
    prep = ctx.prepare_keys(logn, pk=pk)
    assert prep.shape == (count, n)
    acc, gdbg = ctx.verify_many_prepared(logn, prep, msg, sig, debug=True)
    _compare_debug(logn, flags, odbg, acc, gdbg)
    accw = ctx.verify_many_prepared(logn, prep, msg, sig, norm_policy=1)
    np.testing.assert_array_equal(accw, (flags >> 1) & 1)
    # decoded-polynomial input gives the same prepared blob for keys whose header is right
    good = pk[:, 0] == logn
    h = np.stack([oracle.decode_pkey(logn, k)[1] for k in pk[good]])
    np.testing.assert_array_equal(ctx.prepare_keys(logn, h=h), prep[good])
    # a few keys shared by many items: every item is checked against key (i % nkeys), so most fail the norm
    kidx = (np.arange(count) % nkeys).astype(np.uint32)
    want, wdbg = oracle.verify_many(logn, pk[kidx], msg, sig, debug=True, threads=8)
    acc2, gdbg2 = ctx.verify_many_prepared(logn, prep[:nkeys], msg, sig, key_idx=kidx, debug=True)
    _compare_debug(logn, want, wdbg, acc2, gdbg2)
    assert acc2[:nkeys].tolist() == (flags[:nkeys] & 1).tolist()
    # a handful of items: the one-wave-per-item front ahead of the prepared-key kernels, with and without the index
    for m in (1, 7, 200):
        a3, g3 = ctx.verify_many_prepared(logn, prep[:m], msg[:m], sig[:m], debug=True)
        _compare_debug(logn, flags[:m], odbg[:m], a3, g3)
        a4, g4 = ctx.verify_many_prepared(logn, prep[:nkeys], msg[:m], sig[:m], key_idx=kidx[:m], debug=True)
        _compare_debug(logn, want[:m], wdbg[:m], a4, g4)
    with pytest.raises(Exception):
        ctx.verify_many_prepared(logn, prep[:nkeys], msg, sig, key_idx=kidx + 1)  # index out of range is refused


@pytest.mark.parametrize("logn", [9, 10])
def test_hash_to_point_many(oracle, logn):
    """hashing::hash_to_point (hashing.hpp:16-50) as a batched primitive, messages of 0..1024 bytes; a private
    context with too few squeeze blocks exercises the re-hash pass."""
    from falcon_amd import capi
    rng = np.random.default_rng(logn)
    lens = [0, 1, 32, 95, 96, 97, 135, 136, 137, 231, 232, 233, 500, 1024] * 10  # 40 + len crosses 136 at 96, 272 at 232
    salt = rng.integers(0, 256, (len(lens), 40), dtype=np.uint8)
    msgs = [rng.integers(0, 256, m, dtype=np.uint8) for m in lens]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    flat = np.concatenate(msgs)
    want = np.stack([oracle.hash_to_point(logn, salt[i], msgs[i])[0] for i in range(len(lens))])
    c = capi.Context(0)
    try:
        np.testing.assert_array_equal(c.hash_to_point_many(logn, salt, flat, msg_off=off), want)  # one sponge per wave at this size
        c.set_option(capi.OPT_TINY_BATCH_ITEMS, 0)
        np.testing.assert_array_equal(c.hash_to_point_many(logn, salt, flat, msg_off=off), want)  # one sponge per lane
        c.set_option(capi.OPT_HASH_BLOCKS_512 if logn == 9 else capi.OPT_HASH_BLOCKS_1024, 8 if logn == 9 else 16)
        c.set_option(capi.OPT_CHUNK_ITEMS, 64)
        for tiny in (0, 512):
            c.set_option(capi.OPT_TINY_BATCH_ITEMS, tiny)
            c.get_profile()
            np.testing.assert_array_equal(c.hash_to_point_many(logn, salt, flat, msg_off=off), want)
            assert c.get_profile()["retried_items"] > 0
        fixed = np.stack([m for m in msgs if m.size == 32])
        np.testing.assert_array_equal(c.hash_to_point_many(logn, salt[np.array(lens) == 32], fixed), want[np.array(lens) == 32])
    finally:
        c.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_compute_public_keys(ctx, oracle, logn):
    """keygen::compute_public_key (keygen.hpp:67-90) batched: equal to the oracle, including the full coefficient
    range [-6145, 6143] and an f with zeros in the NTT domain (division by zero yields 0)."""
    n = 1 << logn
    rng = np.random.default_rng(7 + logn)
    f = rng.integers(-60, 61, (130, n)).astype(np.int32)
    g = rng.integers(-60, 61, (130, n)).astype(np.int32)
    f[1] = rng.integers(-6145, 6144, n)
    g[1] = rng.integers(-6145, 6144, n)
    f[2] = 0
    f[2, 0] = 1          # f = 1: h = g
    f[3] = 0             # f = 0: every quotient element is 0
    f[4] = 0
    f[4, 0], f[4, n // 2] = 1, 1   # 1 + x^(n/2): zero at half of the NTT points
    h = ctx.compute_public_keys(logn, f, g)
    want = np.stack([oracle.compute_public_key(logn, f[i], g[i]) for i in range(f.shape[0])])
    np.testing.assert_array_equal(h, want)
    assert h[2].tolist() == (g[2] % 12289).tolist() and not h[3].any()
    # h * f == g through the GPU product for the invertible keys
    fq = (f % 12289).astype(np.uint16)
    inv = np.array([not (oracle.ntt(fq[i], logn) == 0).any() for i in range(f.shape[0])])
    prod = ctx.polymul_many(logn, h, fq)
    np.testing.assert_array_equal(prod[inv], (g % 12289).astype(np.uint16)[inv])


@pytest.mark.parametrize("logn", [9, 10])
def test_fuzz_random_bytes(ctx, oracle, corpus, logn):
    """Differential fuzz (the shape of tests/test_encoding.cpp:243-288, which feeds random bytes to the
    decompressor): random keys and signatures with right and wrong headers, sparse random bodies (long zero runs:
    the k >= 8 window rule) and single-bit damage to valid items anywhere in key, salt or body."""
    rng = np.random.default_rng(2024 + logn)
    n = 1500
    pk, msg, sig, _ = corpus.generate(logn, n, seed=808 + logn)
    third = n // 3
    # (a) fully random records, headers right three times out of four
    pk[:third] = rng.integers(0, 256, pk[:third].shape, dtype=np.uint8)
    sig[:third] = rng.integers(0, 256, sig[:third].shape, dtype=np.uint8)
    pk[:third, 0] = np.where(rng.random(third) < 0.75, logn, pk[:third, 0])
    sig[:third, 0] = np.where(rng.random(third) < 0.75, 0x30 | logn, sig[:third, 0])
    # (b) sparse bodies: mostly zero bits, so unary runs of 8 and more occur all the time
    body = (rng.random((third, sig.shape[1] - 41, 8)) < 0.12)
    sig[third:2 * third, 41:] = np.packbits(body, axis=2)[:, :, 0]
    # (c) one flipped bit somewhere in a valid key or signature
    for i in range(2 * third, n):
        if rng.random() < 0.5:
            pk[i, rng.integers(0, pk.shape[1])] ^= 1 << rng.integers(0, 8)
        else:
            sig[i, rng.integers(0, sig.shape[1])] ^= 1 << rng.integers(0, 8)
    flags, odbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    acc, gdbg = ctx.verify_many(logn, pk, msg, sig, debug=True)
    _compare_debug(logn, flags, odbg, acc, gdbg)
    accw = ctx.verify_many(logn, pk, msg, sig, norm_policy=1)
    np.testing.assert_array_equal(accw, (flags >> 1) & 1)
    assert len(set(odbg["status"].tolist())) >= 4  # the fuzz reaches several outcomes


@pytest.mark.parametrize("logn", [9, 10])
def test_host_pipeline_sub_batches(oracle, corpus, logn):
    """fa_verify_many / fa_verify_many_prepared cut a host batch into sub-batches whose copies overlap the previous
    sub-batch's kernels (two staging sets). Small, non-dividing sub-batches exercise every hand-over: ragged message
    offsets (a window of the offset array per sub-batch), the debug outputs, prepared keys with and without an index,
    pageable and pinned caller memory."""
    from falcon_amd import capi
    c = capi.Context(0)
    pinned = []
    try:
        c.set_option(capi.OPT_HOST_CHUNK_ITEMS, 192)
        c.set_option(capi.OPT_CHUNK_ITEMS, 128)
        count = 1000  # 5 full sub-batches and one of 40 items
        pk, _, sig, _ = corpus.generate(logn, count, seed=91 + logn, malformed_permille=300)
        rng = np.random.default_rng(5)
        lens = rng.integers(0, 300, count)
        lens[:3] = (0, 1, 1024)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        flat = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
        flags, odbg = oracle.verify_many(logn, pk, flat, sig, msg_off=off, debug=True, threads=8)
        acc, gdbg = c.verify_many(logn, pk, flat, sig, msg_off=off, debug=True)
        _compare_debug(logn, flags, odbg, acc, gdbg)
        # fixed-length messages, pinned inputs and output
        pk2, msg2, sig2, _ = corpus.generate(logn, count, seed=93 + logn, malformed_permille=200)
        want = oracle.verify_many(logn, pk2, msg2, sig2, threads=8)[0]
        bufs = [capi.PinnedArray(a.shape) for a in (pk2, msg2, sig2)] + [capi.PinnedArray((count,))]
        pinned.extend(bufs)
        for b, a in zip(bufs, (pk2, msg2, sig2)):
            b.array[...] = a
        out = c.verify_many(logn, bufs[0].array, bufs[1].array, bufs[2].array, accept_out=bufs[3].array)
        assert out is bufs[3].array
        np.testing.assert_array_equal(out, want & 1)
        np.testing.assert_array_equal(c.verify_many(logn, pk2, msg2, sig2, norm_policy=1), (want >> 1) & 1)
        # caller memory pinned in place (fa_host_register), then released again
        # (hipHostRegister is refused with "invalid argument" on some hosts, this pool's boxes among them: the call
        # then reports that error and the memory simply stays pageable)
        if c.lib.fa_host_register(sig2.ctypes.data, sig2.nbytes) == 0:
            try:
                np.testing.assert_array_equal(c.verify_many(logn, pk2, msg2, sig2), want & 1)
            finally:
                assert c.lib.fa_host_unregister(sig2.ctypes.data) == 0
        assert c.lib.fa_host_register(None, 16) != 0 and c.lib.fa_host_unregister(None) != 0
        # prepared keys: one per item, then a few shared ones named by index
        prep = c.prepare_keys(logn, pk=pk2)
        np.testing.assert_array_equal(c.verify_many_prepared(logn, prep, msg2, sig2), want & 1)
        kidx = (np.arange(count) % 37).astype(np.uint32)
        want2 = oracle.verify_many(logn, pk2[kidx], msg2, sig2, threads=8)[0]
        np.testing.assert_array_equal(c.verify_many_prepared(logn, prep[:37], msg2, sig2, key_idx=kidx), want2 & 1)
        # sub-batches large enough for the threaded staging of pageable memory (4 MiB pieces), and the runtime's own path
        if logn == 9:
            c.set_option(capi.OPT_HOST_CHUNK_ITEMS, 6000)
            c.set_option(capi.OPT_CHUNK_ITEMS, 1 << 20)
            pk3, msg3, sig3, _ = corpus.generate(logn, 16384, seed=97, malformed_permille=250, threads=8)
            want3 = oracle.verify_many(logn, pk3, msg3, sig3, threads=8)[0] & 1
            for threads in (6, 1, 0):
                c.set_option(capi.OPT_HOST_THREADS, threads)
                np.testing.assert_array_equal(c.verify_many(logn, pk3, msg3, sig3), want3, err_msg=f"host threads {threads}")
    finally:
        for b in pinned:
            b.close()
        c.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_ntt_many(ctx, oracle, logn):
    """ntt::ntt / ntt::intt as batched operators (ntt.hpp:120-200): element for element the reference's arrays
    (bit-reversed output order included), and the round trip of tests/test_ntt.cpp:48-57."""
    n = 1 << logn
    rng = np.random.default_rng(100 + logn)
    a = rng.integers(0, 12289, (70, n)).astype(np.uint16)
    a[0] = np.arange(n) % 12289
    a[1] = 12288
    a[2] = 0
    a[3] = rng.integers(12289, 65536, n)  # non-canonical input: taken mod q, as the ff_t constructor does
    fwd = ctx.ntt_many(logn, a)
    want = np.stack([oracle.ntt(row % 12289, logn) for row in a])
    np.testing.assert_array_equal(fwd, want)
    inv = ctx.ntt_many(logn, fwd, inverse=True)
    np.testing.assert_array_equal(inv, a % 12289)
    b = rng.integers(0, 65536, (5, n)).astype(np.uint16)
    np.testing.assert_array_equal(ctx.ntt_many(logn, b, inverse=True), np.stack([oracle.intt(row % 12289, logn) for row in b]))
    assert ctx.ntt_many(logn, a[:0]).shape == (0, n)


@pytest.mark.parametrize("logn", [9, 10])
def test_decode_pkeys_and_sigs(ctx, oracle, corpus, logn):
    """decoding::decode_pkey and decoding::decode_sig as batched operators, every malformed class included: the
    same h / salt / s2 / bool as the oracle, and rows of failing items written exactly as far as the reference
    writes them (decoding.hpp:29-32, 359, 380-387)."""
    n = 1 << logn
    count = 700
    pk, _, sig, kinds = corpus.generate(logn, count, seed=55 + logn, malformed_permille=500)
    h, ok = ctx.decode_pkeys(logn, pk)
    for i in range(count):
        o_ok, o_h = oracle.decode_pkey(logn, pk[i])
        assert bool(ok[i]) == o_ok
        np.testing.assert_array_equal(h[i], o_h)
    assert (ok == 0).any() and (h >= 12289).any()  # wrong headers and coefficients >= q are both in the corpus
    salt0 = np.full((count, 40), 0xA5, np.uint8)
    s20 = np.full((count, n), 0x5A5A, np.int16)
    salt, s2, sok = ctx.decode_sigs(logn, sig, salt=salt0.copy(), s2=s20.copy())
    seen = set()
    for i in range(count):
        o_ok, o_salt, o_s2 = oracle.decode_sig(logn, sig[i], salt=salt0[i].copy(), s2=s20[i].astype(np.int32))
        assert bool(sok[i]) == o_ok, (i, support.KINDS[kinds[i]])
        np.testing.assert_array_equal(salt[i], o_salt)
        np.testing.assert_array_equal(s2[i].astype(np.int32), o_s2)
        seen.add((o_ok, bool((o_s2 == 0x5A5A).all())))
    assert seen == {(True, False), (False, True), (False, False)}  # decoded, wrong header (untouched), bad body (zeroed)
    # chunked: several passes of the decoders
    from falcon_amd import capi
    c = capi.Context(0)
    try:
        c.set_option(capi.OPT_CHUNK_ITEMS, 128)
        salt_c, s2_c, ok_c = c.decode_sigs(logn, sig, salt=salt0.copy(), s2=s20.copy())
        np.testing.assert_array_equal(salt_c, salt)
        np.testing.assert_array_equal(s2_c, s2)
        np.testing.assert_array_equal(ok_c, sok)
    finally:
        c.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_decoders_on_crafted_bodies(ctx, oracle, logn):
    """Bodies built to stress the code-boundary search of all three decoders (wave-parallel, lane-per-item fast path,
    generic): constant bytes, sparse and dense random bits (long zero runs = k >= 8 windows, decoding.hpp:300-320),
    maximum-length codes up to the last bits, a code ending exactly on the last bit (decoding.hpp:340), one-bit
    perturbations of valid bodies. ok flags and s2 identical to the oracle for every front."""
    from falcon_amd import capi
    n, slen = 1 << logn, support.sig_len(logn)
    rng = np.random.default_rng(1000 + logn)
    body = slen - 41
    sigs = []

    def with_body(b):
        s = np.zeros(slen, np.uint8)
        s[0] = 0x30 | logn
        s[1:41] = rng.integers(0, 256, 40, dtype=np.uint8)
        s[41:] = b
        sigs.append(s)

    for byte in (0x00, 0xFF, 0xAA, 0x55, 0x01, 0x80, 0x7F, 0xFE):
        with_body(np.full(body, byte, np.uint8))
    for density in (0.02, 0.05, 0.1, 0.2, 0.35, 0.5, 0.65, 0.8, 0.95):
        for _ in range(6):
            with_body(np.packbits(rng.random(body * 8) < density).astype(np.uint8))
    budget = 8 * body
    for big in (0, 1, n // 8, n // 4, n // 2):  # `big` coefficients with seven unary zeros each, the rest small
        for _ in range(4):
            s2 = rng.integers(-127, 128, n).astype(np.int32)
            idx = rng.choice(n, big, replace=False)
            s2[idx] = rng.choice([-1, 1], big) * rng.integers(7 * 128, 8 * 128, big)
            ok, sg = oracle.compress_sig(logn, s2)
            with_body(sg[41:])
    # fill the body exactly: small coefficients are 9 bits each, every 128 more in magnitude one bit more
    for slack in (0, 1, 2, 9):
        s2 = np.zeros(n, np.int32)
        extra = budget - 9 * n - slack  # unary zeros to hand out, at most 7 per coefficient
        k = np.minimum(7, np.maximum(0, extra - 7 * np.arange(n)))
        s2 = (k * 128 + 1).astype(np.int32)
        ok, sg = oracle.compress_sig(logn, s2)
        with_body(sg[41:] if ok else np.zeros(body, np.uint8))
    base = [s.copy() for s in sigs[-24:-4]]
    for s in base:  # one flipped bit in a valid body
        t = s.copy()
        bit = int(rng.integers(0, 8 * body))
        t[41 + bit // 8] ^= 0x80 >> (bit % 8)
        sigs.append(t)
    sig = np.stack(sigs)
    want_ok, want_s2 = [], []
    for srow in sig:
        o, _, s2o = oracle.decode_sig(logn, srow)
        want_ok.append(o)
        want_s2.append(s2o if o else np.zeros(n, np.int32))
    want_ok, want_s2 = np.array(want_ok), np.stack(want_s2)
    assert want_ok.any() and not want_ok.all()
    c = capi.Context(0)
    try:
        for tiny, chunk in ((512, 1 << 20), (0, 1 << 20), (512, 64)):  # wave-parallel; fast path + generic fix-up; wave-parallel in several passes
            c.set_option(capi.OPT_TINY_BATCH_ITEMS, tiny)
            c.set_option(capi.OPT_CHUNK_ITEMS, chunk)
            _, s2, ok = c.decode_sigs(logn, sig)
            np.testing.assert_array_equal(ok.astype(bool), want_ok)
            np.testing.assert_array_equal(s2.astype(np.int32), want_s2)
    finally:
        c.close()


@pytest.mark.parametrize("logn", [9, 10])
def test_reference_output_fixture(ctx, logn):
    """tests/golden/ref_outputs.npz: what the reference's own code returns for the inputs of corpus_small.npz
    (tests/golden/make_ref_golden.py). The HIP operators and the verify path's debug outputs must equal them."""
    g = np.load(os.path.join(GOLDEN, "corpus_small.npz"), allow_pickle=False)
    r = np.load(os.path.join(GOLDEN, "ref_outputs.npz"), allow_pickle=False)
    n = 1 << logn
    pk, sig = g[f"pk{logn}"], g[f"sig{logn}"]
    h, ok = ctx.decode_pkeys(logn, pk)
    np.testing.assert_array_equal(ok, r[f"pk_ok{logn}"])
    np.testing.assert_array_equal(h, r[f"h{logn}"])
    salt, s2, sok = ctx.decode_sigs(logn, sig)  # zeroed outputs, as the fixture's calls
    np.testing.assert_array_equal(sok, r[f"sig_ok{logn}"])
    np.testing.assert_array_equal(salt, r[f"salt{logn}"])
    np.testing.assert_array_equal(s2, r[f"s2_{logn}"])
    acc, dbg = ctx.verify_many(logn, pk, g[f"msg{logn}"], sig, msg_off=g[f"off{logn}"], debug=True)
    both = (r[f"pk_ok{logn}"] & r[f"sig_ok{logn}"]).astype(bool)
    np.testing.assert_array_equal(dbg["norm_exact"][both], r[f"norm{logn}"][both])
    np.testing.assert_array_equal(dbg["s1"][both], r[f"s1_{logn}"][both])
    np.testing.assert_array_equal(acc[both], (r[f"norm{logn}"][both] <= support.BETA2[logn]).astype(np.uint8))
    assert not acc[~both].any()
    np.testing.assert_array_equal(ctx.ntt_many(logn, r[f"poly{logn}"]), r[f"ntt{logn}"])
    np.testing.assert_array_equal(ctx.ntt_many(logn, r[f"poly{logn}"], inverse=True), r[f"intt{logn}"])


# ---- the encoders beside the path ----

@pytest.mark.parametrize("logn", [9, 10])
def test_encoders_against_oracle_and_round_trip(ctx, oracle, corpus, logn):
    """fa_encode_pkeys / fa_compress_sigs (encoding.hpp:12-35, 169-230) against the oracle row by row; the decoders undo
    them (tests/test_encoding.cpp:14-48, 104-237); re-encoding what the decoders give back for a corpus reproduces its
    bytes (tests/test_encoding.cpp:243-288)."""
    n = 1 << logn
    rng = np.random.default_rng(40 + logn)
    count = 333                                   # not a multiple of 64: a ragged last wave
    h = rng.integers(0, 12289, (count, n)).astype(np.uint16)
    h[::5] = rng.integers(0, 65536, h[::5].shape).astype(np.uint16)       # out-of-range values: packed as the reference packs them
    pk = ctx.encode_pkeys(logn, h)
    for i in range(count):
        np.testing.assert_array_equal(pk[i], oracle.encode_pkey(logn, h[i]))
    h2, ok = ctx.decode_pkeys(logn, pk[1::5])
    assert ok.all()
    np.testing.assert_array_equal(h2, h[1::5])
    s2 = np.empty((count, n), np.int16)
    for i in range(count):
        s2[i] = np.rint(rng.normal(0, [20, 80, 165, 250, 400][i % 5], n))
    s2[7, :40] = 2047                              # long unary runs
    s2[8, 100] = -32768                            # far outside a signer's range: counted, fails
    s2[9] = 0
    s2[10, 5] = -128
    sig0 = rng.integers(0, 256, (count, support.sig_len(logn)), dtype=np.uint8)
    sig, okc = ctx.compress_sigs(logn, s2, sig0.copy())
    n_ok = 0
    for i in range(count):
        o_ok, o_sig = oracle.compress_sig(logn, s2[i].astype(np.int32), sig0[i].copy())
        assert bool(okc[i]) == o_ok, i
        np.testing.assert_array_equal(sig[i], o_sig)
        n_ok += o_ok
    assert 0 < n_ok < count and not okc[8]
    sig[:, 0] = 0x30 | logn
    salt, back, okd = ctx.decode_sigs(logn, sig)
    # a code that fits decodes to what went in, as long as no unary run reaches 8 zeros (the reference's decoder drops
    # those); a zeroed body does not decode
    plain = (okc == 1) & (np.abs(s2.astype(np.int32)).max(axis=1) < 1024)
    assert plain.sum() > count // 3 and okd[plain].all() and not okd[okc == 0].any()
    np.testing.assert_array_equal(back[plain], s2[plain])
    # corpus items: decode, re-encode, compare bytes
    cpk, _, csig, kinds = corpus.generate(logn, 200, seed=77, malformed_permille=0)
    dh, dok = ctx.decode_pkeys(logn, cpk)
    np.testing.assert_array_equal(ctx.encode_pkeys(logn, dh), cpk)
    dsalt, ds2, dsok = ctx.decode_sigs(logn, csig)
    assert dok.all() and dsok.all()
    re_sig, re_ok = ctx.compress_sigs(logn, ds2, csig.copy())
    assert re_ok.all()
    np.testing.assert_array_equal(re_sig, csig)
