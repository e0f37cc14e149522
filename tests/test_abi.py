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


def test_lateral_boundary_logic_picks_the_cells_the_reference_does():
    """CellBlock.lateral_boundary_conditions (host logic of wrf_pmc_trans_aero.F90:803-955) on a
    stand-in block: which boundary cells copy their inner neighbour, which are resampled, which
    of those are emptied first -- for one block and for a slab of a domain cut in y."""
    import numpy as np
    from types import SimpleNamespace
    from wrf_partmc_amd.host import CellBlock

    class Fake:
        cfg = SimpleNamespace(n_cells=0)
        def __init__(self, n): self.cfg = SimpleNamespace(n_cells=n); self.copied = None; self.res = None
        def boundary_copy(self, f, t): self.copied = (list(f), list(t))
        def boundary_resample(self, cells, thr, bg, conc, dens, ct, seed, step):
            self.res = (list(cells), list(thr)); return np.zeros(len(cells), np.int32), np.zeros(len(cells), np.int32)

    nx, nz, ny = 4, 2, 5
    cell = lambda i, k, j: (j * nz + k) * nx + i
    u_w = np.full((ny, nz), 1.0); u_w[2, :] = -1.0          # row 2 of the west side flows out
    u_e = np.full((ny, nz), -1.0); u_e[1, 0] = 2.0          # one east cell flows out
    v_s = np.full((nx, nz), 1.0); v_n = np.full((nx, nz), 1.0)   # south: inflow, north: outflow
    u_w_old = u_w.copy(); u_w_old[3, 1] = -0.5              # the wind turned there: emptied first
    bg = [dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[1.0])]
    f = Fake(nx * nz * ny)
    out = CellBlock.lateral_boundary_conditions(f, nx, nz, ny, u_w, u_e, v_s, v_n, bg, 1.0, 1.0, u_west_old=u_w_old)
    pairs = set(zip(*f.copied))
    assert {(cell(1, k, 2), cell(0, k, 2)) for k in range(nz)} <= pairs           # west outflow row
    assert (cell(nx - 2, 0, 1), cell(nx - 1, 0, 1)) in pairs                      # the east outflow cell
    assert {(cell(i, k, ny - 2), cell(i, k, ny - 1)) for i in (1, 2) for k in range(nz)} <= pairs   # north
    assert len(pairs) == 2 + 1 + 4
    res, thr = f.res
    assert cell(0, 0, 0) not in res and cell(nx - 1, 1, ny - 1) not in res and cell(0, 0, 0) not in f.copied[1]  # corners
    assert set(res) == ({cell(0, k, j) for j in (1, 3) for k in range(nz)} |
                        {cell(nx - 1, k, j) for j in (1, 2, 3) for k in range(nz)} - {cell(nx - 1, 0, 1)} |
                        {cell(i, k, 0) for i in (1, 2) for k in range(nz)})
    assert [c for c, t in zip(res, thr) if t] == [cell(0, 1, 3)]
    # a slab holding domain rows 2..4 (local rows 1..3, ghosts 0 and 4) of a 5-row domain: its
    # west/east columns on rows 2, 3; the north row (4) is its own; no south row
    g = Fake(nx * nz * 5)
    CellBlock.lateral_boundary_conditions(g, nx, nz, 5, u_w[[1, 2, 3, 4, 0]], u_e[[1, 2, 3, 4, 0]], v_s, v_n, bg, 1.0, 1.0,
                                          j_global0=1, global_ny=ny)
    to = set(g.copied[1]) | set(g.res[0])
    assert to == ({cell(i, k, j) for i in (0, nx - 1) for k in range(nz) for j in (1, 2)} |
                  {cell(i, k, 3) for i in (1, 2) for k in range(nz)})
