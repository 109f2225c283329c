"""The oracle must reproduce the committed fixture (tests/golden/corpus_small.npz, see make_golden.py)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus_small.npz")


def load_golden():
    return np.load(GOLDEN, allow_pickle=False)


@pytest.mark.parametrize("logn", [9, 10])
def test_oracle_reproduces_fixture(oracle, logn):
    g = load_golden()
    n = 1 << logn
    flags, dbg = oracle.verify_many(logn, g[f"pk{logn}"], g[f"msg{logn}"], g[f"sig{logn}"], msg_off=g[f"off{logn}"],
                                    debug=True)
    np.testing.assert_array_equal(flags, g[f"flags{logn}"])
    np.testing.assert_array_equal(dbg["status"], g[f"status{logn}"])
    np.testing.assert_array_equal(dbg["norm_exact"], g[f"norm{logn}"])
    np.testing.assert_array_equal(dbg["c"][:, :n], g[f"c{logn}"])
    np.testing.assert_array_equal(dbg["s1"][:, :n], g[f"s1_{logn}"])
    np.testing.assert_array_equal(dbg["s2"][:, :n], g[f"s2_{logn}"])


REF_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_outputs.npz")


@pytest.mark.parametrize("logn", [9, 10])
def test_oracle_matches_reference_outputs(oracle, logn):
    """tests/golden/ref_outputs.npz holds what the REFERENCE's own code returns for the inputs of corpus_small.npz
    (made by make_ref_golden.py from oracle/_ref where /root/reference exists): decode_pkey, decode_sig, ntt / intt
    and, for items that decode, s1 and the norm. The oracle must agree with every one of them."""
    g, r = load_golden(), np.load(REF_GOLDEN, allow_pickle=False)
    n = 1 << logn
    pk, sig = g[f"pk{logn}"], g[f"sig{logn}"]
    flags, dbg = oracle.verify_many(logn, pk, g[f"msg{logn}"], sig, msg_off=g[f"off{logn}"], debug=True)
    both = 0
    for i in range(pk.shape[0]):
        ok, h = oracle.decode_pkey(logn, pk[i])
        ok2, salt, s2 = oracle.decode_sig(logn, sig[i])
        assert ok == bool(r[f"pk_ok{logn}"][i]) and ok2 == bool(r[f"sig_ok{logn}"][i])
        np.testing.assert_array_equal(h, r[f"h{logn}"][i])
        np.testing.assert_array_equal(salt, r[f"salt{logn}"][i])
        np.testing.assert_array_equal(s2, r[f"s2_{logn}"][i])
        if ok and ok2:
            both += 1
            assert int(dbg["norm_exact"][i]) == int(r[f"norm{logn}"][i])
            np.testing.assert_array_equal(dbg["s1"][i][:n], r[f"s1_{logn}"][i])
    assert both >= 16
    for p, f, b in zip(r[f"poly{logn}"], r[f"ntt{logn}"], r[f"intt{logn}"]):
        np.testing.assert_array_equal(oracle.ntt(p, logn), f)
        np.testing.assert_array_equal(oracle.intt(p, logn), b)
