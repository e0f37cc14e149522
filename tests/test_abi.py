"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950,
loads, and exports every symbol include/pmcx.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pmcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmcx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(wp):
    lib = wp.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libpmcx.so does not export {name}"
    assert sorted(wp.EXPORTED_SYMBOLS) == declared


def test_config_default_matches_reference_defaults(wp):
    lib = wp.load_library()
    cfg = wp.PmcxConfig()
    lib.pmcx_config_default(ctypes.byref(cfg))
    assert cfg.struct_size == ctypes.sizeof(wp.PmcxConfig)
    assert cfg.n_dbin == 100 and cfg.d_min == 1e-9 and cfg.d_max == 1e-3   # driver:783
    assert cfg.n_spec == 20                                                 # aero_data.dat
    assert cfg.n_bin == 80 and cfg.n_class == 1


def test_create_fails_loudly_without_gpu(wp):
    """No CPU fallback: without a gfx950 device pmcx_create returns an error."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    import pytest
    with pytest.raises(wp.PmcxError):
        wp.CellBlock(1, 64, 1)


def test_bad_config_rejected(wp):
    lib = wp.load_library()
    cfg = wp.PmcxConfig()
    lib.pmcx_config_default(ctypes.byref(cfg))
    cfg.n_bin = 1000
    h = ctypes.c_void_p()
    assert lib.pmcx_create(ctypes.byref(cfg), ctypes.byref(h)) == 1   # PMCX_EINVAL
    assert b"out of range" in lib.pmcx_last_error(None)


def test_code_objects_are_gfx950_only(wp):
    """The shared object carries exactly one device target."""
    import subprocess
    so = wp.library_path()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={so}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets)
    else:
        data = open(so, "rb").read()
        assert b"gfx950" in data and b"gfx942" not in data and b"sm_90" not in data
