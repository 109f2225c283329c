"""CPU-side checks of the drop-in boundary: libfalcon_amd.so loads, exports every symbol that
include/falcon_amd.h declares, and rejects bad arguments before touching a GPU. No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "falcon_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from falcon_amd import capi
    lib = capi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/falcon_amd.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_sizes_and_error_strings():
    from falcon_amd import capi
    lib = capi.load_library()
    assert lib.fa_pkey_len(9) == 897 and lib.fa_pkey_len(10) == 1793   # utils.hpp:9-17
    assert lib.fa_sig_len(9) == 666 and lib.fa_sig_len(10) == 1280     # utils.hpp:39-49
    assert lib.fa_pkey_len(8) == 0 and lib.fa_sig_len(11) == 0
    assert lib.fa_strerror(0) == b"ok"
    assert b"logn" in lib.fa_strerror(-2)


def test_null_context_is_rejected():
    from falcon_amd import capi
    lib = capi.load_library()
    assert lib.fa_verify_many(None, 9, 1, None, None, None, 0, None, 0, None, None) == -1
    assert lib.fa_verify_many_dev(None, 9, 1, None, None, None, 0, None, 0, None, None, None) == -1
    assert lib.fa_polymul_many(None, 9, 1, None, None, None) == -1
    assert lib.fa_set_option(None, 1, 64) == -1
    assert lib.fa_ctx_create(0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """There is no CPU fallback: without the HIP library the product raises."""
    from falcon_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libfalcon_amd.so"))
    with pytest.raises(capi.FalconAmdError):
        capi.load_library()


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under falcon_amd/ or include/ may name it."""
    for sub in ("falcon_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "falcon_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_cpp_host_header_compiles():
    """include/falcon_amd.hpp (falcon::verify<N>, verify_many<N>) builds as C++20 against the C ABI."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "host_api_check"))
