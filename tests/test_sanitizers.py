"""CPU-side sanitizer leg: the oracle, the corpus generator and -- where /root/reference exists --
the reference's own decoders and transforms (oracle/ref_harness.cpp) built with -fsanitize=address,undefined and run
over the differential-fuzz cases of tests/test_ref_parity.py::test_decompress_fuzz plus every malformed class of the
corpus. Each record reaches the code under test in a heap block of exactly its size, so the over-read of
decoding.hpp:217-218 would be an AddressSanitizer report if it left the harness's padded copy, and any over-read of
the oracle's restatement likewise. (verification.hpp:76's int32 overflow is not observable: that header needs the
absent sha3 submodule and is not part of oracle/_ref.) Never run on the GPU: host code only."""
import os
import subprocess

import numpy as np
import pytest

import support

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")


def _build(target, exe):
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), target], check=True, capture_output=True, timeout=300)
    except (subprocess.CalledProcessError, OSError) as e:  # no compiler / no sanitizer runtime on this machine
        pytest.skip(f"cannot build {target}: {getattr(e, 'stderr', b'')[-300:]!r}")
    path = os.path.join(SAN, exe)
    if not os.path.exists(path):
        pytest.skip(f"{exe} not built (no /root/reference here)")
    return path


def _cases(corpus, logn):
    """The inputs of test_ref_parity.py::test_decompress_fuzz (bit flips, random and sparse bodies, cut signatures) under
    valid keys, then 40 items of each of the 12 malformed classes and of the valid class."""
    slen = support.sig_len(logn)
    rng = np.random.default_rng(700 + logn)
    vpk, vmsg, valid, _ = corpus.generate(logn, 64, seed=9, malformed_permille=0)
    sigs = []
    for t in range(600):
        s = valid[t % 64].copy()
        for _ in range(1 + t % 4):
            bit = int(rng.integers(8 * 41, 8 * slen))
            s[bit >> 3] ^= 0x80 >> (bit & 7)
        sigs.append(s)
    for t in range(300):
        s = rng.integers(0, 256, slen, dtype=np.uint8)
        if t % 2:
            s[41:] &= rng.integers(0, 256, slen - 41, dtype=np.uint8)
        if t % 3 == 0:
            s[41:] &= rng.integers(0, 256, slen - 41, dtype=np.uint8)
        s[0] = 0x30 | logn
        sigs.append(s)
    for t in range(150):
        s = valid[t % 64].copy()
        cut = int(rng.integers(41, slen))
        s[cut:] = 0 if t % 2 else rng.integers(0, 256, slen - cut, dtype=np.uint8)
        sigs.append(s)
    for tail in (0x00, 0x01, 0x80, 0xFF):  # the last bytes of the stream, where the reference reads ahead
        s = valid[0].copy()
        s[-3:] = tail
        sigs.append(s)
    sig = np.stack(sigs)
    pk = vpk[np.arange(sig.shape[0]) % 64]
    msg = vmsg[np.arange(sig.shape[0]) % 64]
    for kind in range(13):
        p, m, s, _ = corpus.generate(logn, 40, seed=300 + kind, kind=kind)
        pk, msg, sig = np.concatenate([pk, p]), np.concatenate([msg, m]), np.concatenate([sig, s])
    return pk, msg, sig


def _run(cmd):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)  # the sanitizer runtime must come first in the link order
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr and "LeakSanitizer" not in out.stderr, out.stderr[-3000:]
    return out.stdout.strip()


@pytest.mark.parametrize("logn", [9, 10])
def test_oracle_and_corpus_under_sanitizers(tmp_path, oracle, corpus, logn):
    exe = _build("san", "san_driver")
    pk, msg, sig = _cases(corpus, logn)
    for name, a in (("pk", pk), ("msg", msg), ("sig", sig)):
        np.ascontiguousarray(a).tofile(tmp_path / f"{name}.bin")
    digest = _run([exe, str(logn), str(pk.shape[0]), str(msg.shape[1]), str(tmp_path / "pk.bin"), str(tmp_path / "msg.bin"),
                   str(tmp_path / "sig.bin")])
    assert len(digest) == 16
    # the cases are not trivial for the decoder: some decode, most do not
    oks = [oracle.decompress_sig(logn, s)[0] for s in sig[:200]]
    assert 0 < sum(oks) < 200


@pytest.mark.parametrize("logn", [9, 10])
def test_reference_headers_under_sanitizers(tmp_path, corpus, logn):
    if not os.path.isdir("/root/reference/include"):
        pytest.skip("the reference tree is not on this machine")
    exe = _build("san_ref", "san_ref_driver")
    pk, _, sig = _cases(corpus, logn)
    np.ascontiguousarray(pk).tofile(tmp_path / "pk.bin")
    np.ascontiguousarray(sig).tofile(tmp_path / "sig.bin")
    digest = _run([exe, str(logn), str(pk.shape[0]), str(tmp_path / "pk.bin"), str(tmp_path / "sig.bin")])
    assert len(digest) == 16
