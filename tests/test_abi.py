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


# test hooks and diagnostics of include/pmcx.h: everything else is a product entry point
_TEST_HOOKS = {"pmcx_get_bin_tables", "pmcx_get_diag_grid", "pmcx_get_kmax", "pmcx_get_caches", "pmcx_get_sort",
               "pmcx_get_keys", "pmcx_test_math", "pmcx_test_philox", "pmcx_set_sampler_variant",
               "pmcx_profile_phase_cycles", "pmcx_invalidate_lists", "pmcx_get_sampler_info"}


def _struct_fields(header, name):
    body = header[:header.index("} %s;" % name)]
    body = body[body.rindex("typedef struct"):]
    out = []
    for decl in re.findall(r"\b(?:const\s+)?(?:u?int32_t|u?int64_t|double|float|pmcx_aero_dist)\s*\*?\s*([a-z_0-9, *]+);", body):
        out += [f.strip().lstrip("*").strip() for f in decl.split(",")]
    return out


def test_fortran_binds_every_product_entry_point():
    """wrf-partmc_amd/fortran/pmcx_c.F90 binds every function include/pmcx.h declares except the
    test hooks -- a product entry point without a Fortran binding fails here -- names nothing the
    header does not declare, and its bind(c) mirrors list the header's struct fields in order."""
    fdir = os.path.join(ROOT, "wrf-partmc_amd", "fortran")
    iface = open(os.path.join(fdir, "pmcx_c.F90")).read()
    shim = open(os.path.join(fdir, "pmcx_shim.F90")).read()
    bound = set(re.findall(r'bind\(c,\s*name="(pmcx_[a-z0-9_]+)"\)', iface))
    declared = set(_declared_symbols())
    assert bound <= declared, sorted(bound - declared)
    unbound = declared - bound - _TEST_HOOKS
    assert not unbound, f"product entry points without a Fortran binding: {sorted(unbound)}"
    assert 'bind(c' not in shim.replace("bind(c) ::", "")       # the shim declares no interface of its own
    # the two drop-in drivers carry the reference's argument lists
    def args(src, name):
        m = re.search(r"subroutine\s+%s\s*\(([^)]*)\)" % name, src, flags=re.S)
        return [a.strip() for a in re.sub(r"&|\s", "", m.group(1)).split(",")]
    ref_ts = ["grid", "config_flags", "scenario", "env_states", "aero_data", "aero_states", "gas_data", "gas_states",
              "pmc_is", "pmc_ie", "pmc_js", "pmc_je", "pmc_ks", "pmc_ke", "global_nx", "global_ny", "global_nz"]
    ref_tr = ["grid", "config_flags", "env_states", "scenario", "aero_data", "aero_states", "gas_states", "pmc_is",
              "pmc_ie", "pmc_js", "pmc_je", "pmc_ks", "pmc_ke", "global_nx", "global_ny", "global_nz"]
    assert args(shim, "partmc_timestep_pmcx") == ref_ts            # wrf_pmc_driver.F90:46-48
    assert args(shim, "wrf_pmc_trans_driver_pmcx") == ref_tr       # wrf_pmc_trans.F90:37-39
    assert args(shim, "mc_coag_pmcx") == ["coag_kernel_type", "env_state", "aero_data", "aero_state", "del_t",
                                           "tot_n_samp", "tot_n_coag"]                       # driver:192-194
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pmcx.h")).read(), flags=re.S)
    for c_name, f_name in (("pmcx_config", "pmcx_config"), ("pmcx_trans_fields", "pmcx_trans_fields_t"),
                           ("pmcx_aero_dist", "pmcx_aero_dist_t"), ("pmcx_step_flags", "pmcx_step_flags_t")):
        f_type = iface[iface.index("type, bind(c) :: %s\n" % f_name):iface.index("end type %s\n" % f_name)]
        f_fields = [f.strip() for grp in re.findall(r"::\s*([a-z_0-9, ]+)", f_type.split("\n", 1)[1])
                    for f in grp.split(",")]
        assert f_fields == _struct_fields(header, c_name), (c_name, f_fields, _struct_fields(header, c_name))


def test_ctypes_mirrors_match_the_header_layout(wp, tmp_path):
    """The ctypes Structures of host.py against the C compiler's own layout of include/pmcx.h:
    size and every field offset (a caller built against another header must not pass a shorter
    struct unnoticed: pmcx_config and pmcx_step_flags also carry their size)."""
    import subprocess
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pmcx.h")).read(), flags=re.S)
    pairs = (("pmcx_config", wp.PmcxConfig), ("pmcx_trans_fields", wp.host.PmcxTransFields),
             ("pmcx_aero_dist", wp.host.PmcxAeroDist), ("pmcx_step_flags", wp.host.PmcxStepFlags))
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "pmcx.h"\nint main(void){\n'
    for c_name, _ in pairs:
        src += 'printf("%s %%zu", sizeof(%s));\n' % (c_name, c_name)
        for f in _struct_fields(header, c_name):
            src += 'printf(" %s=%%zu", offsetof(%s, %s));\n' % (f, c_name, f)
        src += 'printf("\\n");\n'
    src += "return 0;}\n"
    (tmp_path / "layout.c").write_text(src)
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / "layout"),
                           str(tmp_path / "layout.c")])
    lines = subprocess.check_output([str(tmp_path / "layout")], text=True).strip().split("\n")
    for (c_name, cls), line in zip(pairs, lines):
        parts = line.split()
        assert parts[0] == c_name and int(parts[1]) == ctypes.sizeof(cls), (line, ctypes.sizeof(cls))
        want = [(p.split("=")[0], int(p.split("=")[1])) for p in parts[2:]]
        got = [(n, getattr(cls, n).offset) for n, _ in cls._fields_]
        assert got == want, (c_name, got, want)
    flags = wp.host.PmcxStepFlags()
    wp.load_library().pmcx_step_flags_default(ctypes.byref(flags))
    assert flags.struct_size == ctypes.sizeof(wp.host.PmcxStepFlags) and flags.partmc_chem_dt == 60.0


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
