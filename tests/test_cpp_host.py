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
    out = subprocess.run([EXE, str(logn), str(count), "32", str(tmp_path / "pk.bin"), str(tmp_path / "msg.bin"),
                          str(tmp_path / "sig.bin"), "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    many, single, prepared, sharded = out.stdout.split("\n")[:4]
    want = "".join(str(int(f) & 1) for f in oracle.verify_many(logn, pk, msg, sig)[0])
    assert many == want
    assert single == want[:8]
    assert prepared == want
    assert sharded == want
