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
