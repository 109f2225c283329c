"""GPU parity against the REFERENCE ITSELF (oracle/_ref/libfalcon_ref.so, see tests/test_ref_parity.py): the HIP
operators, called through the C ABI, element for element against the reference's decode_pkey, decode_sig, ntt,
intt, and -- for s1 and the norm of a verify -- against verification.hpp:36-79 run with the reference's functions."""
import numpy as np
import pytest

import support

ref = support.reference()
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(ref is None, reason="oracle/_ref/libfalcon_ref.so not built (needs /root/reference)")]


@pytest.mark.parametrize("logn", [9, 10])
def test_operators_against_reference(ctx, corpus, logn):
    n = 1 << logn
    count = 400
    pk, msg, sig, kinds = corpus.generate(logn, count, seed=500 + logn, malformed_permille=500)
    h, ok = ctx.decode_pkeys(logn, pk)
    salt0 = np.full((count, 40), 0xA5, np.uint8)
    s20 = np.full((count, n), 0x5A5A, np.int16)
    salt, s2, sok = ctx.decode_sigs(logn, sig, salt=salt0.copy(), s2=s20.copy())
    acc, dbg = ctx.verify_many(logn, pk, msg, sig, debug=True)
    checked = 0
    for i in range(count):
        r_ok, r_h = ref.decode_pkey(logn, pk[i])
        assert bool(ok[i]) == r_ok
        np.testing.assert_array_equal(h[i], r_h)
        r_ok2, r_salt, r_s2 = ref.decode_sig(logn, sig[i], salt0[i].copy(), s20[i].astype(np.int32))
        assert bool(sok[i]) == r_ok2, support.KINDS[kinds[i]]
        np.testing.assert_array_equal(salt[i], r_salt)
        np.testing.assert_array_equal(s2[i].astype(np.int32), r_s2)
        if r_ok and r_ok2:
            s1, norm = ref.verify_after_hash(logn, r_h, dbg["c"][i], r_s2)
            assert norm == int(dbg["norm_exact"][i])
            np.testing.assert_array_equal(dbg["s1"][i], s1)
            np.testing.assert_array_equal(dbg["s2"][i].astype(np.int32), r_s2)
            assert bool(acc[i]) == (norm <= support.BETA2[logn])
            checked += 1
        else:
            assert acc[i] == 0
    assert checked > count // 3
    rng = np.random.default_rng(logn)
    a = rng.integers(0, 12289, (48, n)).astype(np.uint16)
    fwd = ctx.ntt_many(logn, a)
    inv = ctx.ntt_many(logn, a, inverse=True)
    for i in range(48):
        np.testing.assert_array_equal(fwd[i], ref.ntt(a[i], logn))
        np.testing.assert_array_equal(inv[i], ref.ntt(a[i], logn, inverse=True))


@pytest.mark.parametrize("logn", [9, 10])
def test_encoders_against_reference(ctx, logn):
    """The HIP encoders against the reference's own encode_pkey / compress_sig (encoding.hpp:12-35, 169-230)."""
    n = 1 << logn
    rng = np.random.default_rng(60 + logn)
    count = 200
    h = rng.integers(0, 12289, (count, n)).astype(np.uint16)
    h[::4] = rng.integers(0, 65536, h[::4].shape).astype(np.uint16)
    s2 = np.empty((count, n), np.int16)
    for i in range(count):
        s2[i] = np.rint(rng.normal(0, [30, 120, 165, 300][i % 4], n))
    pk = ctx.encode_pkeys(logn, h)
    sig, ok = ctx.compress_sigs(logn, s2)
    for i in range(count):
        np.testing.assert_array_equal(pk[i], ref.encode_pkey(logn, h[i]))
        r_ok, r_sig = ref.compress_sig(logn, s2[i].astype(np.int32))
        assert bool(ok[i]) == r_ok
        np.testing.assert_array_equal(sig[i][41:], r_sig[41:])
    assert 0 < ok.sum() < count
