"""CPU: the build-time checks of the device assembly (tools/check_scalar_loads.py, tools/check_asm_loads.py, both run by
falcon_amd/csrc/build.sh)."""
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
    assert "check_scalar_loads.py" in text and "check_asm_loads.py" in text and "set -euo pipefail" in text


# ---- loads written as instructions (verify_kernel.hpp prefetch_s2): nothing names their target before the wait ----
ASM_TOOL = os.path.join(ROOT, "tools", "check_asm_loads.py")

# The shape of the verify kernel's item loop: loads issued in a block that is laid out AFTER the block holding the wait
# (the loop closes backwards), one block on the way that the loads skip.
LOOP = """\
_ZN2fa4loopEv:
\ts_mov_b32 s0, 0
\tglobal_load_ushort v8, v1, s[2:3] ; FA_ASM_LOAD
\ts_waitcnt vmcnt(0) ; FA_ASM_LOAD_WAIT
.LBB0_1:
\tv_add_u32_e32 v9, v8, v8
\ts_cbranch_scc1 .LBB0_3
.LBB0_2:
\ts_waitcnt vmcnt(0) ; FA_ASM_LOAD_WAIT
\t%(latch)s
\ts_cbranch_scc0 .LBB0_1
\ts_endpgm
.LBB0_3:
\tglobal_load_ushort v8, v1, s[2:3] ; FA_ASM_LOAD
\tv_mul_lo_u32 v10, v9, v9
\t%(body)s
\ts_branch .LBB0_2
.Lfunc_end0:
"""


def _asm(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, ASM_TOOL, str(f)], capture_output=True, text=True)


def test_loads_waited_for_along_the_control_flow_pass(tmp_path):
    r = _asm(tmp_path, LOOP % {"latch": "v_mov_b32_e32 v11, v8", "body": "s_nop 0"})
    assert r.returncode == 0 and "2 marked loads, 0 findings" in r.stdout, r.stdout


def test_a_copy_before_the_wait_is_found(tmp_path):
    r = _asm(tmp_path, LOOP % {"latch": "s_nop 0", "body": "v_mov_b32_e32 v11, v8"})
    assert r.returncode == 1 and "v8 named while its load" in r.stdout
    # a register range that covers the target counts too
    r = _asm(tmp_path, LOOP % {"latch": "s_nop 0", "body": "ds_write_b64 v0, v[7:8]"})
    assert r.returncode == 1 and "v8 named while its load" in r.stdout


def test_a_missing_wait_is_found_across_the_back_edge(tmp_path):
    text = (LOOP % {"latch": "s_nop 0", "body": "s_nop 0"}).replace(
        ".LBB0_2:\n\ts_waitcnt vmcnt(0) ; FA_ASM_LOAD_WAIT", ".LBB0_2:\n\ts_nop 1")
    r = _asm(tmp_path, text)   # the load of block 3 is still in flight when block 1 adds v8
    assert r.returncode == 1 and "v_add_u32_e32 v9, v8, v8" in r.stdout


def test_no_marked_load_is_an_error(tmp_path):
    assert _asm(tmp_path, "_ZN2fa1xEv:\n\ts_endpgm\n.Lfunc_end0:\n").returncode == 1
