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


def test_fortran_shim_binds_only_declared_entry_points():
    """Every bind(c, name=...) of the ISO_C_BINDING shim names a function include/pmcx.h
    declares, and the bind(c) mirror of pmcx_config lists the header's fields in order."""
    shim = open(os.path.join(ROOT, "wrf-partmc_amd", "fortran", "pmcx_shim.F90")).read()
    bound = set(re.findall(r'bind\(c,\s*name="(pmcx_[a-z0-9_]+)"\)', shim))
    declared = set(_declared_symbols())
    assert len(bound) >= 20 and bound <= declared, sorted(bound - declared)
    # the path itself and the rows around it are all reachable from Fortran
    for name in ("pmcx_mc_coag", "pmcx_rebalance", "pmcx_transport", "pmcx_dry_deposition",
                 "pmcx_transport_begin", "pmcx_transport_finish", "pmcx_upload_cells",
                 "pmcx_download_cells"):
        assert name in bound, name
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pmcx.h")).read(), flags=re.S)
    struct = header[header.index("typedef struct {"):header.index("} pmcx_config;")]
    c_fields = re.findall(r"\b(?:u?int32_t|double)\s+([a-z_0-9, ]+);", struct)
    c_fields = [f.strip() for grp in c_fields for f in grp.split(",")]
    f_type = shim[shim.index("type, bind(c) :: pmcx_config"):shim.index("end type pmcx_config")]
    f_fields = [f.strip() for grp in re.findall(r"::\s*([a-z_0-9, ]+)", f_type.split("\n", 1)[1])
                for f in grp.split(",")]
    assert f_fields == c_fields, (f_fields, c_fields)


def test_integration_notes_cover_every_entry_point():
    """INTEGRATION.md names the reference interface each C entry point replaces: none is left out."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in _declared_symbols() if s not in doc]
    assert not missing, missing
