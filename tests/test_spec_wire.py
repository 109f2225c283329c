"""FA_WIRE_SPEC: the Falcon specification's wire format beside the reference's (optional half).
The reference has no counterpart (README.md:74-75: it is not wire compatible) and no official known-answer vector is
available offline: PARITY UNPINNED. What is checked: the oracle's restatement of the specification's codecs
against an independent bit-string restatement in Python and against round trips (CPU), and the HIP path against that
oracle on valid items and on every way an encoding can be invalid (GPU)."""
import numpy as np
import pytest

import support

Q = 12289


def py_compress(s2, nbits):
    """Algorithm 17 on a bit string; None when the code does not fit."""
    bits = []
    for v in s2:
        v = int(v)
        mag = abs(v)
        bits.append("1" if v < 0 else "0")
        bits.append(format(mag & 0x7f, "07b"))
        bits.append("0" * (mag >> 7) + "1")
    out = "".join(bits)
    return None if len(out) > nbits else out + "0" * (nbits - len(out))


def py_decompress(bits, n):
    """Algorithm 18 on a bit string; None for an invalid encoding."""
    pos, out = 0, []
    for _ in range(n):
        if pos + 8 > len(bits):
            return None
        sign, low = bits[pos] == "1", int(bits[pos + 1:pos + 8], 2)
        pos += 8
        k = 0
        while True:
            if pos >= len(bits):
                return None
            b = bits[pos]
            pos += 1
            if b == "1":
                break
            k += 1
            if k > 15:
                return None
        mag = low + 128 * k
        if sign and mag == 0:
            return None
        out.append(-mag if sign else mag)
    return None if "1" in bits[pos:] else out


def sig_bits(sig):
    return "".join(format(int(b), "08b") for b in sig[41:])


def bits_to_sig(logn, bits, head):
    sig = np.zeros(support.sig_len(logn), np.uint8)
    sig[:41] = head
    body = np.frombuffer(int(bits, 2).to_bytes(len(bits) // 8, "big"), np.uint8)
    sig[41:] = body
    return sig


def crafted_signatures(logn, rng):
    """(name, signature bytes) for every way a specification encoding can be special or invalid."""
    n, nbits = 1 << logn, 8 * (support.sig_len(logn) - 41)
    head = np.concatenate([[0x30 | logn], rng.integers(0, 256, 40)]).astype(np.uint8)
    base = np.clip(np.rint(rng.normal(0, 165, n)), -600, 600).astype(np.int64)
    out = []
    big = base.copy()
    big[[3, 77, n - 1]] = (1024, -2047, 1500)               # unary runs of 8..15 zeros: valid here, not in the reference
    out.append(("k_8_to_15", bits_to_sig(logn, py_compress(big, nbits), head)))
    ok_bits = py_compress(base, nbits)
    # a code with 16 zeros in its unary part
    code = "0" + format(5, "07b") + "0" * 16 + "1"
    rest = py_compress(base[1:], nbits - len(code))
    out.append(("k_16", bits_to_sig(logn, code + rest, head)))
    # minus zero
    mz = "1" + "0000000" + "1" + py_compress(base[1:], nbits - 9)
    out.append(("minus_zero", bits_to_sig(logn, mz, head)))
    # non-zero padding: last bit set
    out.append(("padding_bit", bits_to_sig(logn, ok_bits[:-1] + "1", head)))
    # the last code ends exactly on the last bit: magnitudes raised until the stream is full
    full = base.copy()
    raw = "".join(("1" if v < 0 else "0") + format(abs(int(v)) & 0x7f, "07b") + "0" * (abs(int(v)) >> 7) + "1" for v in full)
    i = 0
    while len(raw) < nbits:
        step = min(15 - (abs(int(full[i])) >> 7), nbits - len(raw))
        full[i] += (128 * step) * (1 if full[i] >= 0 else -1)
        raw = "".join(("1" if v < 0 else "0") + format(abs(int(v)) & 0x7f, "07b") + "0" * (abs(int(v)) >> 7) + "1" for v in full)
        i += 1
    assert len(raw) == nbits
    out.append(("ends_on_last_bit", bits_to_sig(logn, raw, head)))
    # one bit too long: the last code loses its terminator
    out.append(("unterminated", bits_to_sig(logn, raw[:-1] + "0", head)))
    # wrong header
    bad = bits_to_sig(logn, ok_bits, head)
    bad[0] ^= 0x01
    out.append(("wrong_header", bad))
    out.append(("plain", bits_to_sig(logn, ok_bits, head)))
    return out


@pytest.mark.parametrize("logn", [9, 10])
def test_spec_codecs_against_python_restatement(oracle, logn):
    n, nbits = 1 << logn, 8 * (support.sig_len(logn) - 41)
    rng = np.random.default_rng(logn)
    # public keys: round trip, the packing itself, the range check
    h = rng.integers(0, Q, n).astype(np.uint16)
    pk = oracle.spec_encode_pkey(logn, h)
    want = int("".join(format(int(v), "014b") for v in h), 2).to_bytes(14 * n // 8, "big")
    assert pk[0] == logn and pk[1:].tobytes() == want
    ok, back = oracle.spec_decode_pkey(logn, pk)
    assert ok and (back == h).all()
    h2 = h.copy()
    h2[n // 3] = Q  # not a field element
    ok, back = oracle.spec_decode_pkey(logn, oracle.spec_encode_pkey(logn, h2))
    assert not ok and not back.any()
    bad = pk.copy()
    bad[0] ^= 0x10
    assert not oracle.spec_decode_pkey(logn, bad)[0]
    # signatures: crafted cases and fuzz, C against Python
    cases = [s for _, s in crafted_signatures(logn, rng)]
    for t in range(400):
        s = cases[t % 8].copy()
        for _ in range(1 + t % 3):
            bit = int(rng.integers(8 * 41, 8 * support.sig_len(logn)))
            s[bit >> 3] ^= 0x80 >> (bit & 7)
        cases.append(s)
    n_ok = 0
    for s in cases:
        want = py_decompress(sig_bits(s), n)
        ok, got = oracle.spec_decompress_sig(logn, s)
        assert ok == (want is not None)
        if ok:
            n_ok += 1
            assert got.tolist() == want
            fits, enc = oracle.spec_compress_sig(logn, got)
            assert fits and (enc[41:] == s[41:]).all()  # the encoding is unique
        else:
            assert not got.any()
    assert 0 < n_ok < len(cases)
    named = dict((k, py_decompress(sig_bits(s), n) is not None) for k, s in crafted_signatures(logn, np.random.default_rng(logn)))
    assert named == {"k_8_to_15": True, "k_16": False, "minus_zero": False, "padding_bit": False, "ends_on_last_bit": True,
                     "unterminated": False, "wrong_header": True, "plain": True}  # the header is not decompress's business


@pytest.mark.parametrize("logn", [9, 10])
def test_spec_verify_of_reencoded_corpus(oracle, corpus, logn):
    """Valid reference-format items re-encoded in the specification's format verify there too, with the same norm
    (no s1 coefficient of a valid item sits at 6144); malformed classes stay rejected."""
    pk, msg, sig, kinds = corpus.generate(logn, 200, seed=70 + logn, malformed_permille=400)
    spk, ssig = support.to_spec_format(logn, pk, sig, oracle)
    ref_flags, ref_dbg = oracle.verify_many(logn, pk, msg, sig, debug=True, threads=8)
    flags, dbg = oracle.verify_many(logn, spk, msg, ssig, debug=True, threads=8, wire=1)
    valid = kinds == 0
    assert (flags[valid] & 1).all() and (dbg["norm_exact"][valid] == ref_dbg["norm_exact"][valid]).all()
    ge_q = kinds == support.KINDS.index("pk_coeff_ge_q")
    assert (ref_flags[ge_q] & 1).all() and not flags[ge_q].any()  # valid for the reference, not a key for the specification


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [9, 10])
def test_gpu_spec_wire_format(oracle, corpus, logn):
    from falcon_amd import capi
    n = 1 << logn
    rng = np.random.default_rng(900 + logn)
    pk, msg, sig, kinds = corpus.generate(logn, 500, seed=80 + logn, malformed_permille=400)
    spk, ssig = support.to_spec_format(logn, pk, sig, oracle)
    crafted = crafted_signatures(logn, rng)
    extra = len(crafted)
    spk = np.concatenate([spk, spk[:extra]])
    msg = np.concatenate([msg, msg[:extra]])
    ssig = np.concatenate([ssig, np.stack([s for _, s in crafted])])
    c = capi.Context(0)
    try:
        c.set_option(capi.OPT_WIRE_FORMAT, capi.WIRE_SPEC)
        for chunk, blocks in ((1 << 20, None), (96, 8 if logn == 9 else 16)):  # one pass; many passes and the re-hash pass
            c.set_option(capi.OPT_CHUNK_ITEMS, chunk)
            if blocks:
                c.set_option(capi.OPT_HASH_BLOCKS_512 if logn == 9 else capi.OPT_HASH_BLOCKS_1024, blocks)
            flags, odbg = oracle.verify_many(logn, spk, msg, ssig, debug=True, threads=8, wire=1)
            acc, gdbg = c.verify_many(logn, spk, msg, ssig, debug=True)
            np.testing.assert_array_equal(acc, flags & 1)
            np.testing.assert_array_equal(gdbg["status"], odbg["status"])
            np.testing.assert_array_equal(gdbg["norm_exact"], odbg["norm_exact"])
            okrow = odbg["status"] >= support.ST_NORM_TOO_LARGE
            okrow |= odbg["status"] == support.ST_OK
            np.testing.assert_array_equal(gdbg["s1"][okrow], odbg["s1"][okrow][:, :n])
            np.testing.assert_array_equal(gdbg["s2"][okrow], odbg["s2"][okrow][:, :n])
            np.testing.assert_array_equal(c.verify_many(logn, spk, msg, ssig, norm_policy=1), (flags >> 1) & 1)
        assert acc[:500][kinds == 0].all()
        by_name = dict(zip([k for k, _ in crafted], odbg["status"][-extra:]))
        assert by_name["k_8_to_15"] != support.ST_BAD_ENCODING and by_name["ends_on_last_bit"] != support.ST_BAD_ENCODING
        assert by_name["k_16"] == by_name["minus_zero"] == by_name["padding_bit"] == by_name["unterminated"] == support.ST_BAD_ENCODING
        assert by_name["wrong_header"] == support.ST_BAD_SIG_HEADER
        # the operators of their own, and prepared keys, in this format
        h, ok = c.decode_pkeys(logn, spk)
        salt, s2, sok = c.decode_sigs(logn, ssig)
        for i in range(spk.shape[0]):
            o_ok, o_h = oracle.spec_decode_pkey(logn, spk[i])
            assert bool(ok[i]) == o_ok and (h[i] == o_h).all()
            o_ok2, o_s2 = oracle.spec_decompress_sig(logn, ssig[i])
            o_ok2 = o_ok2 and ssig[i][0] == (0x30 | logn)
            assert bool(sok[i]) == o_ok2
            if o_ok2:
                assert (s2[i] == o_s2).all() and (salt[i] == ssig[i][1:41]).all()
        prep = c.prepare_keys(logn, pk=spk)
        np.testing.assert_array_equal(c.verify_many_prepared(logn, prep, msg, ssig), flags & 1)
        # the same bytes under the reference's rules give the reference's answers: the option, not the data, decides
        c.set_option(capi.OPT_WIRE_FORMAT, capi.WIRE_REFERENCE)
        np.testing.assert_array_equal(c.verify_many(logn, spk, msg, ssig), oracle.verify_many(logn, spk, msg, ssig, threads=8)[0] & 1)
    finally:
        c.close()
