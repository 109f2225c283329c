"""The timed build of the oracle (oracle/Makefile target `native`, what bench.py's cpu_baseline runs) against the
portable checker build: same source, FO_NATIVE switches the sponge to whole-lane absorb / squeeze and the permutation to
its register form. Both must agree bit for bit, and the native sponge is pinned to hashlib like the portable one."""
import hashlib

import numpy as np
import pytest

import support


@pytest.fixture(scope="module")
def native():
    n = support.oracle_native()
    if n is None:
        pytest.skip("no compiler for the native build")
    return n


def test_native_shake256_matches_hashlib(native):
    data = bytes((i * 7 + 3) & 0xFF for i in range(1000))
    for ln in (0, 1, 7, 8, 9, 40, 72, 135, 136, 137, 271, 272, 273, 999):
        for out in (1, 8, 135, 136, 137, 700):
            assert native.shake256(data[:ln], out) == hashlib.shake_256(data[:ln]).digest(out), (ln, out)


@pytest.mark.parametrize("logn", [9, 10])
def test_native_verify_matches_portable(native, oracle, corpus, logn):
    for mlen in (0, 1, 32, 95, 96, 137, 500):
        pk, msg, sig, _ = corpus.generate(logn, 60, seed=3 + mlen, mlen=mlen, malformed_permille=300)
        f0, d0 = oracle.verify_many(logn, pk, msg, sig, debug=True)
        f1, d1 = native.verify_many(logn, pk, msg, sig, debug=True, threads=2)
        np.testing.assert_array_equal(f0, f1)
        for k in ("c", "s2", "s1", "norm_exact", "norm_wrap32", "status", "hash_blocks"):
            np.testing.assert_array_equal(d0[k], d1[k])
