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
