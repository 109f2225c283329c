"""GPU: the C++20 host API (include/falcon_amd.hpp) through the prebuilt tests/cpp/host_api_check."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_check")


@pytest.mark.parametrize("logn", [9, 10])
def test_cpp_verify_and_verify_many(tmp_path, oracle, corpus, logn):
    assert os.path.exists(EXE), "build tests/cpp first (__graft_entry__.build())"
    count = 300
    pk, msg, sig, _ = corpus.generate(logn, count, seed=77 + logn, malformed_permille=350)
    for name, a in (("pk", pk), ("msg", msg), ("sig", sig)):
        np.ascontiguousarray(a).tofile(tmp_path / f"{name}.bin")
    outdir = tmp_path / "out"
    outdir.mkdir()
    out = subprocess.run([EXE, str(logn), str(count), "32", str(tmp_path / "pk.bin"), str(tmp_path / "msg.bin"),
                          str(tmp_path / "sig.bin"), "1", str(outdir)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    print("\n".join(l for l in out.stderr.split("\n") if "per call" in l))  # the C++ single-call latency, for the log (-s)
    many, single, prepared, sharded, decoded_keys, decoded_single = out.stdout.split("\n")[:6]
    want = "".join(str(int(f) & 1) for f in oracle.verify_many(logn, pk, msg, sig)[0])
    assert many == want
    assert single == want[:8]
    assert prepared == want
    assert sharded == want

    # the operators of the path through their C++ names, against the oracle's functions item by item
    n = 1 << logn
    load = lambda name, dt, shape: np.fromfile(outdir / name, dt).reshape(shape)
    h, hok = load("h.bin", np.uint16, (count, n)), load("hok.bin", np.uint8, count)
    salt, s2, sok = load("salt.bin", np.uint8, (count, 40)), load("s2.bin", np.int16, (count, n)), load("sok.bin", np.uint8, count)
    c, fwd, back = load("c.bin", np.uint16, (count, n)), load("ntt.bin", np.uint16, (count, n)), load("intt.bin", np.uint16, (count, n))
    zero_key_accepts = []
    for i in range(count):
        ok, hh = oracle.decode_pkey(logn, pk[i])
        assert bool(hok[i]) == ok
        np.testing.assert_array_equal(h[i], hh)
        # what the reference leaves untouched stays as the caller initialised it (0xEE salt bytes, 0x5555 coefficients)
        sk, ss, s2o = oracle.decode_sig(logn, sig[i], salt=np.full(40, 0xEE, np.uint8), s2=np.full(n, 0x5555, np.int32))
        assert bool(sok[i]) == sk
        np.testing.assert_array_equal(salt[i], ss)
        np.testing.assert_array_equal(s2[i].astype(np.int32), s2o)
        np.testing.assert_array_equal(c[i], oracle.hash_to_point(logn, salt[i], msg[i])[0])
        np.testing.assert_array_equal(fwd[i], oracle.ntt(h[i], logn))
        np.testing.assert_array_equal(back[i], h[i] % 12289)
        zero_key_accepts.append(want[i] if ok else None)
    # keys prepared from the decoded polynomials: same flags wherever the key's header was right
    for i, w in enumerate(zero_key_accepts):
        if w is not None:
            assert decoded_keys[i] == w, i
    assert decoded_single == decoded_keys[:8]   # verify<N>(h, msg, mlen, sig), one call per item


def test_both_headers_in_one_translation_unit(tmp_path):
    """include/falcon_gpu.hpp beside a falcon::verify<N> with the reference's signature (and, where they were present at
    build time, the reference's own decoding.hpp / ntt.hpp / utils.hpp): compiles, links, verifies the README triple."""
    import json
    exe = os.path.join(ROOT, "tests", "cpp", "both_headers_check")
    assert os.path.exists(exe), "build tests/cpp first (__graft_entry__.build())"
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "readme_kat_falcon512.json")))
    for name in ("pkey", "msg", "sig"):
        (tmp_path / name).write_bytes(bytes.fromhex(k[name]))
    out = subprocess.run([exe, str(tmp_path / "pkey"), str(tmp_path / "msg"), str(tmp_path / "sig")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "verified 1 decode_pkey_agrees 1" in out.stdout
