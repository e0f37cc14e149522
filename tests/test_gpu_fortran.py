"""-m gpu: the Fortran ISO_C_BINDING shim (wrf-partmc_amd/fortran/pmcx_shim.F90),
driven the way partmc_timestep drives mc_coag, against the CPU oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "wrf-partmc_amd", "fortran")
EXE = os.path.join(FDIR, "test_shim")


def _build():
    if not os.path.exists(EXE) and shutil.which("make") and os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        subprocess.call(["make", "-C", FDIR, "-s"])
    if not os.path.exists(EXE):
        pytest.skip("flang or the prebuilt test_shim is not available")


def _check_lists(out, off, ncomp, cm, sl):
    """The component(:) lists test_shim wrote behind a cell's particles against the device's
    (CellBlock.download_components of the same sequence): every source_id and create_time."""
    lists = np.frombuffer(out, np.dtype([("s", "<i4"), ("t", "<f8")]), int(ncomp.sum()), off)
    at = np.concatenate([[0], np.cumsum(ncomp)])
    assert np.array_equal(ncomp, cm["comp_len"][sl])       # (no particle of more than 8 constituents here)
    for i, k in enumerate(ncomp):
        assert np.array_equal(lists["s"][at[i]:at[i + 1]] - 1, cm["comp_source"][sl][i, :k])
        assert np.array_equal(lists["t"][at[i]:at[i + 1]], cm["comp_time"][sl][i, :k])
    return off + 12 * int(ncomp.sum())


@pytest.mark.parametrize("mode", [1, 0])
def test_fortran_shim_matches_oracle(tmp_path, wp, oracle, mode):
    _build()
    n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps = 3, 900, 2, 2, 3, 4, 2
    seed, del_t = 12345, 900.0
    names, dens = parity.default_species(n_spec)
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    # the product's tables are what both sides must use
    with wp.CellBlock(1, 64, n_spec) as blk:
        ve, vs = blk.bin_tables()
        const_k, beta = blk.cfg.const_k, blk.cfg.additive_beta
        max_events, max_appends = blk.cfg.max_events, blk.cfg.max_appends
    modes, num_conc = parity.make_modes(n_spec, n_class)
    cells = []
    # ideal counts that make the first rebalance halve every class (n > 2 n_ideal)
    n_ideal = np.full(n_class, 0.45 * n_part / n_class)
    buf = struct.pack("<7i", n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps)
    buf += struct.pack("<q", seed) + struct.pack("<d", del_t) + dens.astype("<f8").tobytes()
    for c in range(n_cells):
        T, p = 288.15 - 5 * c, 101325.0 - 8000 * c
        oc = oracle.Cell(n_spec, n_class, n_source, 1024, dens, ve, T, p,
                         max_events=max_events, max_appends=max_appends, n_comp_max=8)
        oc.mag[:] = num_conc / n_part * (1 + 0.5 * np.arange(n_class))
        oc.generate(modes, n_part, seed, c, first_id=c << 32, create_time=float(c))
        # every particle its own create time: a coagulated particle's list shows whose entries it holds
        oc.lct[:n_part] = oc.gct[:n_part] = oc.comp_time[:n_part, 0] = c + 1e-4 * np.arange(n_part)
        oc.kmax[:] = oracle.kmax_table(kernel, vs, dens.min(), dens.max(), T, p, const_k, beta)
        cells.append(oc)
        buf += struct.pack("<2d", T, p) + oc.mag.astype("<f8").tobytes() + n_ideal.astype("<f8").tobytes()
        for i in range(n_part):
            buf += oc.vol[:, i].astype("<f8").tobytes() + oc.n_orig[:, i].astype("<i4").tobytes()
            buf += struct.pack("<i", int(oc.wclass[i]) + 1) + struct.pack("<q", int(oc.id[i]))
            buf += struct.pack("<2d", oc.lct[i], oc.gct[i])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(buf)
    r = subprocess.run([EXE, str(fin), str(fout), str(mode)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = fout.read_bytes()
    off = 0
    merged = [set() for _ in cells]          # ids whose composition a coagulation changed
    halved = 0
    for step in range(n_steps):
        ns = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
        nc = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
        for c, oc in enumerate(cells):
            # the shim uploads the Fortran aero_state before every call: caches are
            # recomputed from the species and the sort is not valid (a host-side edit)
            oc.refresh_caches()
            oc.sort_valid[0] = 0
            n0 = int(oc.n_part[0])
            before = dict(zip(oc.id[:n0].tolist(), oc.vol[:, :n0].T.tolist()))
            st, ons, onc = oc.mc_coag(kernel, del_t, seed, c, step, const_k, beta)
            assert (int(ns[c]), int(nc[c])) == (ons, onc), (step, c)
            n1 = int(oc.n_part[0])
            for pid, v in zip(oc.id[:n1].tolist(), oc.vol[:, :n1].T.tolist()):
                if before[pid] != v:
                    merged[c].add(pid)
            # aero_state_rebalance_pmcx: upload (with n_part_ideal), rebalance, download
            oc.refresh_caches()
            oc.sort_valid[0] = 0
            oc.rebalance(n_ideal, True, True, seed, c, step, 1)
            halved += n1 - int(oc.n_part[0])
    assert halved > 0 and all(merged)
    rec = 8 * n_spec + 4 * n_source + 4 + 8 + 16 + 4 + 8 + 8 + 4 + 16
    for c, oc in enumerate(cells):
        n = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
        assert n == int(oc.n_part[0])
        blob = np.frombuffer(out, np.uint8, rec * n, off).reshape(n, rec); off += rec * n
        o = 8 * n_spec
        vol = blob[:, :o].copy().view("<f8")
        cnt = blob[:, o:o + 4 * n_source].copy().view("<i4"); o += 4 * n_source
        cls = blob[:, o:o + 4].copy().view("<i4")[:, 0]
        ids = blob[:, o + 4:o + 12].copy().view("<i8")[:, 0]
        ct = blob[:, o + 12:o + 28].copy().view("<f8"); o += 28
        hyst = blob[:, o:o + 4].copy().view("<i4")[:, 0]
        core = blob[:, o + 4:o + 12].copy().view("<f8")[:, 0]
        absorb = blob[:, o + 12:o + 20].copy().view("<f8")[:, 0]
        ncomp = blob[:, o + 20:o + 24].copy().view("<i4")[:, 0]
        tlim = blob[:, o + 24:o + 40].copy().view("<f8")
        assert np.array_equal(vol.T, oc.vol[:, :n])
        assert np.array_equal(ids, oc.id[:n])
        assert np.array_equal(cls - 1, oc.wclass[:n])
        assert np.array_equal(cnt.T, oc.n_orig[:, :n])          # component(:) <-> source histogram
        assert np.array_equal(ct[:, 0], oc.lct[:n]) and np.array_equal(ct[:, 1], oc.gct[:n])
        # what the device does not carry: re-attached to the particle with the same id, reset
        # where a coagulation changed the particle, never left in a slot that changed hands
        assert np.array_equal(hyst, ids % 7 + 1)
        was_merged = np.array([int(i) in merged[c] for i in ids])
        assert was_merged.any() and not was_merged.all()
        assert np.array_equal(core, np.where(was_merged, 0.0, ids.astype(float)))
        assert np.array_equal(absorb, np.where(was_merged, 0.0, ids.astype(float)))
        assert np.array_equal(ncomp, cnt.sum(1))
        assert (tlim[:, 0] >= oc.lct[:n]).all() and (tlim[:, 1] <= oc.gct[:n]).all()
        assert np.array_equal(tlim[~was_merged, 0], oc.lct[:n][~was_merged])
        # component(:) itself (driver:1421-1437): source_id and create_time of EVERY constituent, in
        # the order the coagulations put them, as the oracle's lists have them
        lists = np.frombuffer(out, np.dtype([("s", "<i4"), ("t", "<f8")]), int(ncomp.sum()), off)
        off += 12 * int(ncomp.sum())
        assert ncomp.max() <= 8 and (ncomp[was_merged] >= 2).all()
        at = np.concatenate([[0], np.cumsum(ncomp)])
        for i in range(n):
            k = int(ncomp[i])
            assert k == oc.comp_n[i], (c, i)
            assert np.array_equal(lists["s"][at[i]:at[i + 1]] - 1, oc.comp_src[i, :k]), (c, i)
            assert np.array_equal(lists["t"][at[i]:at[i + 1]], oc.comp_time[i, :k]), (c, i)
    assert off == len(out)


def test_fortran_resident_sequence(tmp_path, wp):
    """The device-resident calls of the shim (pmcx_upload_states once, then per step
    trans_cells_pmcx, dry_dep_cells_pmcx, mc_coag_resident_pmcx, one pmcx_download_states)
    against the same sequence through the Python host: the Fortran bindings pass every
    argument the way the C-ABI means it.  (Each piece has its own oracle parity test.)"""
    _build()
    n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps = 8, 600, 2, 2, 2, 4, 3
    seed, del_t = 2024, 600.0
    nx, nz, ny = 2, 2, n_cells // 4
    names, dens = parity.default_species(n_spec)
    rs = np.random.RandomState(5)
    cap = (2 * n_part + 63) // 64 * 64
    T = 288.15 - 3.0 * np.arange(n_cells)
    P = 101325.0 - 4000.0 * np.arange(n_cells)
    mag = (1e9 / n_part) * (1 + 0.5 * np.arange(n_class))[None, :] * np.ones((n_cells, 1))
    r = 10 ** rs.uniform(-8.0, -6.0, (n_cells, n_part))
    frac = rs.uniform(0.1, 0.9, (n_cells, n_part))
    vtot = 4.0 / 3.0 * np.pi * r ** 3
    vol = np.stack([vtot * frac, vtot * (1 - frac)], axis=1)              # [cell][spec][part]
    wcl = rs.randint(0, n_class, (n_cells, n_part)).astype(np.int32)
    src = rs.randint(0, n_source, (n_cells, n_part))
    n_orig = np.zeros((n_cells, n_source, n_part), np.int32)
    for s in range(n_source):
        n_orig[:, s, :] = src == s
    ids = (np.arange(n_cells, dtype=np.int64)[:, None] << 32) + np.arange(1, n_part + 1)
    lct = np.tile(np.arange(n_part) * 0.5, (n_cells, 1))
    buf = struct.pack("<7i", n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps)
    buf += struct.pack("<q", seed) + struct.pack("<d", del_t) + dens.astype("<f8").tobytes()
    n_ideal = np.full(n_class, 0.38 * n_part / n_class)    # the first rebalance halves (with room: transport and deposition change the counts first)
    for c in range(n_cells):
        buf += struct.pack("<2d", T[c], P[c]) + mag[c].astype("<f8").tobytes() + n_ideal.astype("<f8").tobytes()
        for i in range(n_part):
            buf += vol[c, :, i].astype("<f8").tobytes() + n_orig[c, :, i].astype("<i4").tobytes()
            buf += struct.pack("<i", int(wcl[c, i]) + 1) + struct.pack("<q", int(ids[c, i]))
            buf += struct.pack("<2d", lct[c, i], lct[c, i] + 1.0)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(buf)
    run = subprocess.run([EXE, str(fin), str(fout), "2"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    out = fout.read_bytes()

    # the same sequence through the Python host
    c0 = np.arange(n_cells)
    lev = (c0 // nx) % nz
    prob_h = 0.02 + 0.005 * np.arange(4)[None, None, :] + 0.001 * np.arange(n_class)[None, :, None]
    prob_h = np.broadcast_to(prob_h, (n_cells, n_class, 4))
    prob_v = np.where(np.arange(nz)[None, None, :] == lev[:, None, None], 0.0, 0.01)
    prob_v = np.broadcast_to(prob_v, (n_cells, n_class, nz))
    with wp.CellBlock(n_cells, cap, n_spec, n_class=n_class, n_source=n_source, n_comp_max=8) as blk:
        blk.set_aero_data(dens)
        blk.set_env(T, P, np.full(n_cells, 0.5))
        blk.set_weights(mag)
        blk.set_n_part_ideal(np.tile(n_ideal, (n_cells, 1)))
        # the shim gives every cell its own block of 2**32 ids of PartMC's counter (pmcx_upload_states)
        blk.set_next_id(1 + (np.arange(n_cells, dtype=np.int64) << 32))
        po = np.arange(n_cells + 1, dtype=np.int64) * n_part
        blk.upload_cells(0, po, vol.transpose(1, 0, 2).reshape(n_spec, -1), wcl.reshape(-1),
                         ids.reshape(-1), lct.reshape(-1), lct.reshape(-1) + 1.0,
                         n_orig.transpose(1, 0, 2).reshape(n_source, -1))
        off = 0
        for step in range(n_steps):
            ex, npw = blk.transport_preweight(nx, nz, ny, prob_h, prob_v, np.ones(n_cells), seed=seed, step=step)
            nm, na = blk.transport(nx, nz, ny, prob_h, prob_v, np.ones(n_cells), seed=seed, step=step)
            nr, bad = blk.dry_deposition(del_t, 30.0 + c0 + 1, 0.3, 1.2, 2e-3, 40.0, 0.8, seed=seed,
                                         step=step, surface=(lev == 0).astype(np.int32))
            ns, nc, st = blk.mc_coag(kernel, del_t, seed, step)
            assert not (st & 3).any()
            blk.aero_state_rebalance(True, True, seed, step)
            if step == 0:
                # the ideal counts are far below the populations: the preweight scaled every weight
                # up (and thinned the particles) before anything moved
                assert npw.sum() > 0.4 * n_cells * n_part
                assert blk.aero_state_total_particles().sum() < 0.7 * n_cells * n_part
            for want in (npw, nm, na, nr, ns, nc):
                got = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
                assert np.array_equal(got, want), step
            got = np.frombuffer(out, "<f8", n_cells * n_class, off).reshape(n_cells, n_class); off += 8 * n_cells * n_class
            assert np.array_equal(got, ex), step          # trans_aero_preweight's expected_num_conc
            assert nm.sum() > 0 and nr.sum() > 0 and nc.sum() > 0
        d = blk.download_cells()
        cm = blk.download_components()
        w_dev = blk.get_weights()
    assert (w_dev > mag * 1.5).all()                       # the scaling raised the magnitudes
    rec = 8 * n_spec + 4 * n_source + 4 + 8 + 16 + 4 + 8 + 8 + 4 + 16
    po = d["part_offset"]
    for c in range(n_cells):
        n = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
        assert n == po[c + 1] - po[c]
        blob = np.frombuffer(out, np.uint8, rec * n, off).reshape(n, rec); off += rec * n
        sl = slice(po[c], po[c + 1])
        o = 8 * n_spec
        assert np.array_equal(blob[:, :o].copy().view("<f8").T, d["vol"][:, sl])
        assert np.array_equal(blob[:, o:o + 4 * n_source].copy().view("<i4").T, d["n_orig"][:, sl])
        o += 4 * n_source
        assert np.array_equal(blob[:, o:o + 4].copy().view("<i4")[:, 0] - 1, d["wclass"][sl])
        assert np.array_equal(blob[:, o + 4:o + 12].copy().view("<i8")[:, 0], d["id"][sl])
        assert np.array_equal(blob[:, o + 12:o + 28].copy().view("<f8"),
                              np.stack([d["least_ct"][sl], d["greatest_ct"][sl]], axis=1))
        off = _check_lists(out, off, blob[:, o + 48:o + 52].copy().view("<i4")[:, 0], cm, sl)
    assert off == len(out)


def test_fortran_resident_sequence_from_fields(tmp_path, wp):
    """The shim's trans_from_fields_cells_pmcx (probabilities from WRF-shaped fields, preweight and
    sampling on the device: only the fields cross the boundary) and seasalt_cells_pmcx, then
    mc_coag and rebalance, against the same sequence through the Python host: counters per step
    and every downloaded particle."""
    _build()
    from tests import trans_fields as tf
    n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps = 8, 600, 2, 2, 2, 4, 3
    seed, del_t = 77, 600.0
    nx, nz, ny = 2, 2, n_cells // 4
    names, dens = parity.default_species(n_spec)
    rs = np.random.RandomState(6)
    cap = (2 * n_part + 63) // 64 * 64
    T = 288.15 - 3.0 * np.arange(n_cells)
    P = 101325.0 - 4000.0 * np.arange(n_cells)
    mag = (1e9 / n_part) * (1 + 0.5 * np.arange(n_class))[None, :] * np.ones((n_cells, 1))
    r = 10 ** rs.uniform(-8.0, -6.0, (n_cells, n_part))
    frac = rs.uniform(0.1, 0.9, (n_cells, n_part))
    vtot = 4.0 / 3.0 * np.pi * r ** 3
    vol = np.stack([vtot * frac, vtot * (1 - frac)], axis=1)
    wcl = rs.randint(0, n_class, (n_cells, n_part)).astype(np.int32)
    src = rs.randint(0, n_source, (n_cells, n_part))
    n_orig = np.zeros((n_cells, n_source, n_part), np.int32)
    for s in range(n_source):
        n_orig[:, s, :] = src == s
    ids = (np.arange(n_cells, dtype=np.int64)[:, None] << 32) + np.arange(1, n_part + 1)
    lct = np.tile(np.arange(n_part) * 0.5, (n_cells, 1))
    n_ideal = np.full(n_class, 0.6 * n_part / n_class)
    buf = struct.pack("<7i", n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps)
    buf += struct.pack("<q", seed) + struct.pack("<d", del_t) + dens.astype("<f8").tobytes()
    for c in range(n_cells):
        buf += struct.pack("<2d", T[c], P[c]) + mag[c].astype("<f8").tobytes() + n_ideal.astype("<f8").tobytes()
        for i in range(n_part):
            buf += vol[c, :, i].astype("<f8").tobytes() + n_orig[c, :, i].astype("<i4").tobytes()
            buf += struct.pack("<i", int(wcl[c, i]) + 1) + struct.pack("<q", int(ids[c, i]))
            buf += struct.pack("<2d", lct[c, i], lct[c, i] + 1.0)
    F = tf.make_fields(nx, nz, ny, n_class, seed=2, k_scale=4.0, flux_scale=0.03 * 0.5 * n_part * mag[0, 0])
    dt, dx, dy = 30.0, 3000.0, 3000.0
    fb = struct.pack("<3d", dt, dx, dy)
    for name in ("alt", "xkhh", "u_flux", "v_flux", "w_flux", "z", "z_at_w", "exch_h"):
        fb += np.ascontiguousarray(F[name], dtype="<f4").tobytes()
    fin, fout, ffld = tmp_path / "in.bin", tmp_path / "out.bin", tmp_path / "fields.bin"
    fin.write_bytes(buf); ffld.write_bytes(fb)
    run = subprocess.run([EXE, str(fin), str(fout), "3", str(ffld)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    out = fout.read_bytes()

    lev = (np.arange(n_cells) // nx) % nz
    with wp.CellBlock(n_cells, cap, n_spec, n_class=n_class, n_source=n_source, n_comp_max=8) as blk:
        blk.set_aero_data(dens)
        blk.set_env(T, P, np.full(n_cells, 0.5))
        blk.set_weights(mag)
        blk.set_n_part_ideal(np.tile(n_ideal, (n_cells, 1)))
        blk.set_next_id(1 + (np.arange(n_cells, dtype=np.int64) << 32))
        po = np.arange(n_cells + 1, dtype=np.int64) * n_part
        blk.upload_cells(0, po, vol.transpose(1, 0, 2).reshape(n_spec, -1), wcl.reshape(-1),
                         ids.reshape(-1), lct.reshape(-1), lct.reshape(-1) + 1.0,
                         n_orig.transpose(1, 0, 2).reshape(n_source, -1))
        off = 0
        tot_emit = 0
        for step in range(n_steps):
            blk.transport_probs(nx, nz, ny, dt, dx, dy, F, download=False)
            blk.transport_preweight(nx, nz, ny, None, None, np.ones(n_cells), seed=seed, step=step)
            nm, na = blk.transport(nx, nz, ny, None, None, np.ones(n_cells), seed=seed, step=step)
            ne, _ = blk.seasalt_emissions(del_t, del_t * step, 1, 0, 1, (lev == 0).astype(np.int32),
                                          6.0 + 0.5 * (np.arange(n_cells) + 1), 40.0, seed=seed, step=step,
                                          mode_class=[0, 1], mode_source=[0, 1])
            ns, nc, st = blk.mc_coag(kernel, del_t, seed, step)
            assert not (st & 3).any()
            blk.aero_state_rebalance(True, True, seed, step)
            for want in (nm, na, ne, ns, nc):
                got = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
                assert np.array_equal(got, want), step
            tot_emit += int(ne.sum())
            assert nm.sum() > 0 and nc.sum() > 0
        assert tot_emit > 0
        d = blk.download_cells()
        cm = blk.download_components()
    rec = 8 * n_spec + 4 * n_source + 4 + 8 + 16 + 4 + 8 + 8 + 4 + 16
    po = d["part_offset"]
    for c in range(n_cells):
        n = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
        assert n == po[c + 1] - po[c]
        blob = np.frombuffer(out, np.uint8, rec * n, off).reshape(n, rec); off += rec * n
        sl = slice(po[c], po[c + 1])
        o = 8 * n_spec
        assert np.array_equal(blob[:, :o].copy().view("<f8").T, d["vol"][:, sl])
        o += 4 * n_source
        assert np.array_equal(blob[:, o + 4:o + 12].copy().view("<i8")[:, 0], d["id"][sl])
        off = _check_lists(out, off, blob[:, o + 48:o + 52].copy().view("<i4")[:, 0], cm, sl)
    assert off == len(out)


def test_transport_across_two_ranks():
    """The N > 1 form of the transport: two processes (one slab each, sharing this GPU, gloo
    rendezvous on 127.0.0.1) exchange their boundary rows' movers with
    CellBlock.transport_distributed; every rank checks its rows against the whole-domain
    transport bit for bit (tests/dist_transport_worker.py)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PMCX_TEST_BACKEND="gloo")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29533",
         os.path.join(ROOT, "tests", "dist_transport_worker.py")],
        capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "distributed transport ok: 2 ranks" in out.stdout


def test_transport_across_a_grid_of_ranks():
    """The second cut direction (the reference decomposes in x and y, wrf_pmc_trans_aero.F90:148-273):
    four processes, a 2 x 2 grid of blocks with ghost rows AND ghost columns, exchange preweight terms,
    weights and movers with CellBlock.transport_preweight_distributed_2d / .transport_distributed_2d
    (gloo rendezvous on 127.0.0.1, one shared GPU); every rank checks its real cells against the
    whole-domain preweight and transport bit for bit, doubly periodic and with open lateral
    boundaries (tests/dist_transport2d_worker.py)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PMCX_TEST_BACKEND="gloo")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
         "--master-addr", "127.0.0.1", "--master-port", "29537",
         os.path.join(ROOT, "tests", "dist_transport2d_worker.py")],
        capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "dist_transport2d_worker ok: 4 ranks" in out.stdout
