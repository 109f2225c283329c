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
