"""CPU: the build-time check of the device assembly (tools/check_scalar_loads.py, run by falcon_amd/csrc/build.sh)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_scalar_loads.py")

GOOD = """\
_ZN2fa4goodEv:
\ts_load_dwordx4 s[8:11], s[2:3], 0x28
\ts_load_dword s4, s[0:1], 0x0
\ts_load_dwordx2 s[6:7], s[0:1], s8          ; register offset: not an immediate
\tv_add_u32_e32 v0, v1, v2
"""
BAD = """\
_ZN2fa3badEv:
\ts_load_dwordx2 s[6:7], s[0:1], 0x10
\ts_load_dwordx4 s[16:19], s[4:5], 0x29      ; the shape hipcc produced for the generic decoder (tiny_kernel.hpp)
"""


def _run(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)


def test_aligned_scalar_loads_pass(tmp_path):
    r = _run(tmp_path, GOOD)
    assert r.returncode == 0 and "3 scalar loads, 0 with" in r.stdout


def test_unaligned_immediate_is_found(tmp_path):
    r = _run(tmp_path, BAD)
    assert r.returncode == 1 and "_ZN2fa3badEv" in r.stderr and "0x29" in r.stderr


def test_no_assembly_is_an_error(tmp_path):
    assert _run(tmp_path, "nothing here\n").returncode == 1


def test_build_script_runs_the_check():
    text = open(os.path.join(ROOT, "falcon_amd", "csrc", "build.sh")).read()
    assert "check_scalar_loads.py" in text and "set -euo pipefail" in text
