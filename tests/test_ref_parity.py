"""The oracle against the REFERENCE ITSELF: oracle/_ref/libfalcon_ref.so is the reference's own headers
(decoding.hpp, encoding.hpp, ntt.hpp, ff.hpp, polynomial.hpp, included where they lie under /root/reference)
behind C entry points (oracle/ref_harness.cpp). hashing.hpp / verification.hpp cannot be built (they need the
absent sha3 submodule), so SHAKE256 stays pinned by hashlib (tests/test_oracle.py) and the steps of
verification::verify after the hash run here through the reference's ntt / polynomial functions.

Skipped where the library has not been built (it is built by `make -C oracle ref` in the container that holds
/root/reference and travels to the GPU box as a built file)."""
import json
import os

import numpy as np
import pytest

import support

Q = 12289
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ref = support.reference()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libfalcon_ref.so not built (needs /root/reference)")


def test_readme_known_answer_through_reference_code(oracle):
    """README.md:308-312 decoded and checked by the reference's own functions; only c comes from the oracle's
    hash_to_point (pinned by hashlib)."""
    k = json.load(open(os.path.join(GOLDEN, "readme_kat_falcon512.json")))
    pk, msg, sig = (np.frombuffer(bytes.fromhex(k[x]), np.uint8) for x in ("pkey", "msg", "sig"))
    ok, h = ref.decode_pkey(9, pk)
    ok2, salt, s2 = ref.decode_sig(9, sig)
    assert ok and ok2
    c, _ = oracle.hash_to_point(9, salt, msg)
    s1, norm = ref.verify_after_hash(9, h, c, s2)
    rec = k["survey_recorded"]
    assert norm == rec["norm_exact"] and norm <= support.BETA2[9]
    assert int((s2.astype(np.int64) ** 2).sum()) == rec["s2_sqnorm"] and s2[:4].tolist() == rec["s2_first4"]
    flags, d = oracle.verify(9, pk, msg, sig)
    np.testing.assert_array_equal(d["s1"][:512], s1)
    np.testing.assert_array_equal(d["s2"][:512], s2)


@pytest.mark.parametrize("logn", [9, 10])
def test_field_ops(logn):
    """ff.hpp:85-142 (+, -, *, unary -) against plain modular arithmetic, which the oracle and the kernels use."""
    rng = np.random.default_rng(logn)
    a = np.concatenate([rng.integers(0, Q, 200000), [0, 0, 1, Q - 1, Q - 1, 6144, 6145]]).astype(np.uint16)
    b = np.concatenate([rng.integers(0, Q, 200000), [0, Q - 1, Q - 1, Q - 1, 1, 6145, 6144]]).astype(np.uint16)
    a64, b64 = a.astype(np.int64), b.astype(np.int64)
    np.testing.assert_array_equal(ref.ff_op(0, a, b), (a64 + b64) % Q)
    np.testing.assert_array_equal(ref.ff_op(1, a, b), (a64 - b64) % Q)
    np.testing.assert_array_equal(ref.ff_op(2, a, b), (a64 * b64) % Q)
    np.testing.assert_array_equal(ref.ff_op(3, a, b), (-a64) % Q)


@pytest.mark.parametrize("logn", [9, 10])
def test_ntt_intt(oracle, logn):
    """ntt::ntt / ntt::intt (ntt.hpp:120-200): the oracle's arrays equal the reference's, element for element."""
    n = 1 << logn
    rng = np.random.default_rng(20 + logn)
    polys = [np.arange(n) % Q, np.full(n, Q - 1), np.zeros(n, np.int64), np.eye(1, n, 0)[0], np.eye(1, n, n - 1)[0]]
    polys += [rng.integers(0, Q, n) for _ in range(40)]
    for p in polys:
        p = p.astype(np.uint16)
        f = ref.ntt(p, logn)
        np.testing.assert_array_equal(oracle.ntt(p, logn), f)
        np.testing.assert_array_equal(oracle.intt(p, logn), ref.ntt(p, logn, inverse=True))
        np.testing.assert_array_equal(ref.ntt(f, logn, inverse=True), p)


@pytest.mark.parametrize("logn", [9, 10])
def test_decoders_on_every_item_class(oracle, corpus, logn):
    """decode_pkey / decode_sig / decompress_sig of the reference against the oracle on the synthetic corpus (valid
    items and the 12 malformed classes), including what a failing call leaves in its outputs; then, for items that
    decode, s1 and the norm through the reference's ntt / polynomial functions."""
    n = 1 << logn
    count = 600
    pk, msg, sig, kinds = corpus.generate(logn, count, seed=300 + logn, malformed_permille=600)
    assert len(set(kinds.tolist())) == len(support.KINDS)
    flags, dbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    checked = 0
    for i in range(count):
        r_ok, r_h = ref.decode_pkey(logn, pk[i])
        o_ok, o_h = oracle.decode_pkey(logn, pk[i])
        assert r_ok == o_ok
        np.testing.assert_array_equal(r_h, o_h)
        salt0, s20 = np.full(40, 0xA5, np.uint8), np.full(n, 0x5A5A, np.int32)
        r_ok2, r_salt, r_s2 = ref.decode_sig(logn, sig[i], salt0.copy(), s20.copy())
        o_ok2, o_salt, o_s2 = oracle.decode_sig(logn, sig[i], salt0.copy(), s20.copy())
        assert r_ok2 == o_ok2, support.KINDS[kinds[i]]
        np.testing.assert_array_equal(r_salt, o_salt)
        np.testing.assert_array_equal(r_s2, o_s2)
        r3, s3 = ref.decompress_sig(logn, sig[i])
        o3, t3 = oracle.decompress_sig(logn, sig[i])
        assert r3 == o3
        np.testing.assert_array_equal(s3, t3)
        if r_ok and r_ok2:
            s1, norm = ref.verify_after_hash(logn, r_h, dbg["c"][i][:n], r_s2)
            assert norm == int(dbg["norm_exact"][i]), support.KINDS[kinds[i]]
            np.testing.assert_array_equal(s1, dbg["s1"][i][:n])
            assert (norm <= support.BETA2[logn]) == bool(flags[i] & 1)
            assert (np.int32(np.uint32(norm & 0xFFFFFFFF)) <= support.BETA2[logn]) == bool(flags[i] & 2)
            checked += 1
        else:
            assert flags[i] == 0
    assert checked > count // 2


@pytest.mark.parametrize("logn", [9, 10])
def test_decompress_fuzz(oracle, corpus, logn):
    """Differential fuzz of decompress_sig (decoding.hpp:267-361): random bodies, bodies with long zero runs (the
    k >= 8 window rule), and single / multiple bit flips of valid signatures."""
    n = 1 << logn
    slen = support.sig_len(logn)
    rng = np.random.default_rng(700 + logn)
    _, _, valid, _ = corpus.generate(logn, 64, seed=9, malformed_permille=0)
    cases = []
    for t in range(1500):
        s = valid[t % 64].copy()
        for _ in range(1 + t % 4):
            bit = int(rng.integers(8 * 41, 8 * slen))
            s[bit >> 3] ^= 0x80 >> (bit & 7)
        cases.append(s)
    for t in range(700):
        s = rng.integers(0, 256, slen, dtype=np.uint8)
        if t % 2:
            s[41:] &= rng.integers(0, 256, slen - 41, dtype=np.uint8)  # sparser ones: longer unary runs
        if t % 3 == 0:
            s[41:] &= rng.integers(0, 256, slen - 41, dtype=np.uint8)
        s[0] = 0x30 | logn
        cases.append(s)
    for t in range(300):  # valid prefix, garbage or zeros after a random cut
        s = valid[t % 64].copy()
        cut = int(rng.integers(41, slen))
        s[cut:] = 0 if t % 2 else rng.integers(0, 256, slen - cut, dtype=np.uint8)
        cases.append(s)
    n_ok = 0
    for s in cases:
        r_ok, r_s = ref.decompress_sig(logn, s)
        o_ok, o_s = oracle.decompress_sig(logn, s)
        assert r_ok == o_ok
        np.testing.assert_array_equal(r_s, o_s)
        n_ok += r_ok
    assert 0 < n_ok < len(cases)


@pytest.mark.parametrize("logn", [9, 10])
def test_corpus_encoders_match_reference_encoders(corpus, logn):
    """The corpus generator restates encode_pkey / compress_sig (encoding.hpp:12-35, 169-230); for valid items the
    reference's own encoders reproduce its bytes from the decoded values (the shape of
    tests/test_encoding.cpp:14-48, 243-288)."""
    pk, _, sig, kinds = corpus.generate(logn, 120, seed=41, malformed_permille=0)
    for i in range(120):
        ok, h = ref.decode_pkey(logn, pk[i])
        assert ok
        np.testing.assert_array_equal(ref.encode_pkey(logn, h), pk[i])
        ok, s2 = ref.decompress_sig(logn, sig[i])
        assert ok
        ok, enc = ref.compress_sig(logn, s2)
        assert ok
        np.testing.assert_array_equal(enc[41:], sig[i][41:])


@pytest.mark.parametrize("logn", [9, 10])
def test_oracle_encoders_match_reference_encoders(oracle, logn):
    """fo_encode_pkey / fo_compress_sig (encoding.hpp:12-35, 169-230) against the reference's own functions: coefficients in
    and out of range, codes that fit, codes that do not, and what a failing compress_sig leaves in the body."""
    n = 1 << logn
    rng = np.random.default_rng(900 + logn)
    for t in range(400):
        h = rng.integers(0, 65536 if t % 3 == 0 else 12289, n).astype(np.uint16)   # values >= 2^14 spill as the reference's casts do
        np.testing.assert_array_equal(oracle.encode_pkey(logn, h), ref.encode_pkey(logn, h))
        sigma = [20, 80, 165, 250, 400][t % 5]
        s2 = np.rint(rng.normal(0, sigma, n)).astype(np.int32)
        if t % 7 == 0:
            s2[rng.integers(0, n, 3)] = rng.integers(-2047, 2048, 3)
        o_ok, o_sig = oracle.compress_sig(logn, s2, np.full(support.sig_len(logn), 0xC3, np.uint8))
        r_ok, r_sig = ref.compress_sig(logn, s2)
        assert o_ok == r_ok
        np.testing.assert_array_equal(o_sig[41:], r_sig[41:])
        assert (o_sig[:41] == 0xC3).all()            # header and salt are the caller's (encoding.hpp:164-165)
        if o_ok and np.abs(s2).max() < 1024:          # tests/test_encoding.cpp:104-237: decompress(compress(s)) == s (the
            # reference's decoder drops unary runs of 8 or more zeros, SURVEY.md 8 a5, so only for |s| < 1024)
            d_ok, d = oracle.decompress_sig(logn, np.concatenate([np.zeros(41, np.uint8), o_sig[41:]]))
            assert d_ok
            np.testing.assert_array_equal(d, s2)
