"""-m gpu: the reference's 20-species table (WRFV3/test/em_scm_xy/aero_data.dat:2-21) on the path.

BASELINE configs[2]: independent cells x 10^4 particles, condensation sub-step (extension) +
Brownian coagulation, em_scm_xy bimodal initial distribution and OC:SO4:NH4 composition
(aero_init_dist.dat, aero_init_comp.dat), RH in U(0.3, 0.95) -- here 64 cells, every particle
field against the oracle bit for bit after each piece, and the partmc_process observables.
BASELINE configs[3]: one column of 30 levels x 10^4 particles driven from Fortran through the
ISO_C_BINDING shim (upload / mc_coag / download per step, then aero_state_rebalance on the host
states), against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import bench
from tests import parity
from wrf_partmc_amd.data import aero_data_arrays, scm_init_modes

pytestmark = pytest.mark.gpu
K_BROWN = 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_shape_twenty_species(wp, oracle):
    n_cells, n_part, cap = 64, 10000, 10048
    dens, kappa, _, names = aero_data_arrays()
    n_spec, i_water = len(dens), names.index("H2O")
    assert n_spec == 20
    modes, num_conc = scm_init_modes()
    blk = wp.CellBlock(n_cells, cap, n_spec, n_class=1, n_source=0, i_water=i_water)
    blk.set_aero_data(dens, kappa=kappa)
    temp, pres = bench.level_profile(30)
    lev = np.arange(n_cells) % 30
    rh = np.random.RandomState(0).uniform(0.3, 0.95, n_cells)
    blk.set_env(temp[lev], pres[lev], rh)
    blk.set_weights(np.full((n_cells, 1), num_conc / n_part))
    blk.set_n_part_ideal(np.full((n_cells, 1), float(n_part)))
    blk.generate_lognormal(n_part, modes, seed=3)
    ve, vs = blk.bin_tables()
    de, dw = blk.diag_grid()
    cells, kcache = [], {}
    for c in range(n_cells):
        oc = oracle.Cell(n_spec, 1, 0, cap, dens, ve, temp[lev[c]], pres[lev[c]],
                         max_events=blk.cfg.max_events, max_appends=blk.cfg.max_appends)
        oc.mag[:] = num_conc / n_part
        oc.generate(modes, n_part, 3, c, first_id=c << 32)
        if lev[c] not in kcache:
            kcache[lev[c]] = oracle.kmax_table(K_BROWN, vs, dens.min(), dens.max(), temp[lev[c]], pres[lev[c]])
        oc.kmax[:] = kcache[lev[c]]
        oc.cell_id = c
        cells.append(oc)
    ok, msg = parity.compare_state(blk, cells)
    assert ok, f"generated population: {msg}"
    events = 0
    for step in range(2):
        blk.condense(300.0, 4)                       # extension: water uptake sub-step
        for c, oc in enumerate(cells):
            oc.condense(kappa, i_water, rh[c], 300.0, 4)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, f"step {step} after condense: {msg}"
        ns, nc, st = blk.mc_coag(K_BROWN, 300.0, 3, step)
        for c, oc in enumerate(cells):
            assert oc.mc_coag(K_BROWN, 300.0, 3, c, step) == (int(st[c]), int(ns[c]), int(nc[c])), (step, c)
        events += int(nc.sum())
        ok, msg = parity.compare_state(blk, cells)
        assert ok, f"step {step} after mc_coag: {msg}"
    assert events > 10 * n_cells
    # the partmc_process observables of the 20-species population (driver:838-932)
    tn, tm, nd, md = blk.diagnostics()
    sm, sn, pm, npart = blk.diagnostics_ext()
    for c in (0, 17, 63):
        otn, otm, ond, omd = cells[c].diagnostics(i_water, de, dw)
        # (the device counts particles per bin and converts once, the oracle adds term by term:
        # the same bins, sums equal to rounding)
        np.testing.assert_allclose(tn[c], otn, rtol=1e-13)
        np.testing.assert_allclose(nd[c], ond, rtol=1e-13)
        assert np.array_equal(nd[c] > 0, ond > 0)
        np.testing.assert_allclose(tm[c], otm, rtol=parity.DIAM_TOL)
        np.testing.assert_allclose(md[c], omd, rtol=1e-11, atol=1e-30)
        osm, _, opm = cells[c].diagnostics_ext(i_water)
        np.testing.assert_allclose(sm[c], osm, rtol=1e-11, atol=1e-30)
        np.testing.assert_allclose(pm[c], opm, rtol=1e-11, atol=1e-30)
        n = int(cells[c].n_part[0])
        # size-distribution L2 (north_star: < 1e-6)
        assert np.linalg.norm(nd[c] - ond) <= 1e-6 * np.linalg.norm(ond)
        assert np.linalg.norm(md[c] - omd) <= 1e-6 * np.linalg.norm(omd)
        assert npart[c] == n
    assert sm[:, names.index("H2O")].min() > 0.0     # every cell took up water
    blk.close()


def test_config3_at_its_stated_size(wp, oracle):
    """BASELINE configs[2] at its stated size: 1000 independent cells x 10^4 particles x 20 species,
    condensation sub-step (extension) + Brownian coagulation, two steps.  Properties on ALL cells --
    every dry species' number-weighted volume conserved by coagulation to rounding, one particle
    lost per event, water taken up, no status bit, ids unique -- and every particle field against
    the oracle, bit for bit, on 64 cells spread over the block."""
    n_cells, n_part, cap = 1000, 10000, 10048
    dens, kappa, _, names = aero_data_arrays()
    n_spec, i_water = len(dens), names.index("H2O")
    modes, num_conc = scm_init_modes()
    blk = wp.CellBlock(n_cells, cap, n_spec, n_class=1, n_source=0, i_water=i_water)
    blk.set_aero_data(dens, kappa=kappa)
    temp, pres = bench.level_profile(30)
    lev = np.arange(n_cells) % 30
    rh = np.random.RandomState(0).uniform(0.3, 0.95, n_cells)
    blk.set_env(temp[lev], pres[lev], rh)
    blk.set_weights(np.full((n_cells, 1), num_conc / n_part))
    blk.set_n_part_ideal(np.full((n_cells, 1), float(n_part)))
    blk.generate_lognormal(n_part, modes, seed=3)
    ve, vs = blk.bin_tables()
    chk = sorted(set(np.linspace(0, n_cells - 1, 64).astype(int).tolist()))
    cells, kcache = {}, {}
    for c in chk:
        oc = oracle.Cell(n_spec, 1, 0, cap, dens, ve, temp[lev[c]], pres[lev[c]],
                         max_events=blk.cfg.max_events, max_appends=blk.cfg.max_appends)
        oc.mag[:] = num_conc / n_part
        oc.generate(modes, n_part, 3, c, first_id=c << 32)
        if lev[c] not in kcache:
            kcache[lev[c]] = oracle.kmax_table(K_BROWN, vs, dens.min(), dens.max(), temp[lev[c]], pres[lev[c]])
        oc.kmax[:] = kcache[lev[c]]
        cells[c] = oc
    sm0 = blk.diagnostics_ext()[0]                                  # species mass concentrations [cell][spec]
    events = np.zeros(n_cells, np.int64)
    dry = [s for s in range(n_spec) if s != i_water]
    for step in range(2):
        blk.condense(300.0, 4)
        sm_c = blk.diagnostics_ext()[0]
        np.testing.assert_allclose(sm_c[:, dry], sm0[:, dry], rtol=1e-12, atol=1e-30)   # condensation moves water only
        n_before = blk.aero_state_total_particles().astype(np.int64)
        ns, nc, st = blk.mc_coag(K_BROWN, 300.0, 3, step)
        assert not st.any()
        n_after = blk.aero_state_total_particles().astype(np.int64)
        assert np.array_equal(n_before - n_after, nc)               # one class: one particle lost per event
        sm1 = blk.diagnostics_ext()[0]
        np.testing.assert_allclose(sm1, sm_c, rtol=1e-11, atol=1e-30)   # coagulation conserves every species
        events += nc
        for c, oc in cells.items():
            oc.condense(kappa, i_water, rh[c], 300.0, 4)
            assert oc.mc_coag(K_BROWN, 300.0, 3, c, step) == (int(st[c]), int(ns[c]), int(nc[c])), (step, c)
        sm0 = sm1
    assert events.min() > 0 and events.sum() > 10 * n_cells
    assert sm0[:, i_water].min() > 0.0                               # every cell took up water
    for c, oc in cells.items():
        dn = blk.download_cells(c, 1)
        n = int(oc.n_part[0])
        assert dn["part_offset"][1] == n
        assert np.array_equal(dn["vol"], oc.vol[:, :n]) and np.array_equal(dn["id"], oc.id[:n]), c
        assert np.array_equal(dn["least_ct"], oc.lct[:n]), c
    ids = blk.download_cells(0, 40)["id"]
    assert len(np.unique(ids)) == len(ids)
    blk.close()


def test_config4_shape_column_through_the_fortran_shim(tmp_path, wp, oracle):
    """30 levels x 10^4 particles x 20 species: Fortran driver -> pmcx_shim -> libpmcx, batched."""
    from tests.test_gpu_fortran import EXE, _build
    _build()
    n_cells, n_part, n_steps, kernel, seed, del_t = 30, 10000, 2, K_BROWN, 0, 300.0
    dens, kappa, _, names = aero_data_arrays()
    n_spec, n_class, n_source = len(dens), 1, 1
    with wp.CellBlock(1, 64, n_spec) as b0:
        ve, vs = b0.bin_tables()
        const_k, beta = b0.cfg.const_k, b0.cfg.additive_beta
        max_events, max_appends = b0.cfg.max_events, b0.cfg.max_appends
    modes, num_conc = scm_init_modes()
    temp, pres = bench.level_profile(n_cells)
    n_ideal = np.full(n_class, float(n_part))
    cells = []
    parts = [struct.pack("<7i", n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps),
             struct.pack("<q", seed), struct.pack("<d", del_t), dens.astype("<f8").tobytes()]
    for c in range(n_cells):
        oc = oracle.Cell(n_spec, n_class, n_source, 20096, dens, ve, temp[c], pres[c],
                         max_events=max_events, max_appends=max_appends)
        oc.mag[:] = num_conc / n_part
        oc.generate(modes, n_part, seed, c, first_id=c << 32)
        oc.kmax[:] = oracle.kmax_table(kernel, vs, dens.min(), dens.max(), temp[c], pres[c], const_k, beta)
        cells.append(oc)
        rec = np.zeros(n_part, dtype=[("vol", "<f8", n_spec), ("no", "<i4", n_source), ("cl", "<i4"),
                                      ("id", "<i8"), ("lct", "<f8"), ("gct", "<f8")])
        rec["vol"] = oc.vol[:, :n_part].T; rec["no"] = oc.n_orig[:, :n_part].T
        rec["cl"] = oc.wclass[:n_part] + 1; rec["id"] = oc.id[:n_part]
        rec["lct"] = oc.lct[:n_part]; rec["gct"] = oc.gct[:n_part]
        parts += [struct.pack("<2d", temp[c], pres[c]), oc.mag.astype("<f8").tobytes(),
                  n_ideal.astype("<f8").tobytes(), rec.tobytes()]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(b"".join(parts))
    r = subprocess.run([EXE, str(fin), str(fout), "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = fout.read_bytes()
    off, events = 0, 0
    for step in range(n_steps):
        ns = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
        nc = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
        events += int(nc.sum())
        for c, oc in enumerate(cells):
            oc.refresh_caches()            # the shim uploads the Fortran view before every call
            oc.sort_valid[0] = 0
            st, ons, onc = oc.mc_coag(kernel, del_t, seed, c, step, const_k, beta)
            assert (int(ns[c]), int(nc[c])) == (ons, onc), (step, c)
            oc.refresh_caches()
            oc.sort_valid[0] = 0
            oc.rebalance(n_ideal, True, True, seed, c, step, 1)
    assert events > 30 * 20
    rec_b = 8 * n_spec + 4 * n_source + 4 + 8 + 16 + 4 + 8 + 8 + 4 + 16
    for c, oc in enumerate(cells):
        n = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
        assert n == int(oc.n_part[0])
        blob = np.frombuffer(out, np.uint8, rec_b * n, off).reshape(n, rec_b); off += rec_b * n
        vol = blob[:, :8 * n_spec].copy().view("<f8")
        o = 8 * n_spec + 4 * n_source
        ids = blob[:, o + 4:o + 12].copy().view("<i8")[:, 0]
        assert np.array_equal(vol.T, oc.vol[:, :n]), c
        assert np.array_equal(ids, oc.id[:n]), c
        ncomp = blob[:, o + 48:o + 52].copy().view("<i4")[:, 0]     # the component(:) lists follow the cell
        assert np.array_equal(ncomp, oc.n_orig[:, :n].sum(0)), c
        off += 12 * int(ncomp.sum())
    assert off == len(out)
