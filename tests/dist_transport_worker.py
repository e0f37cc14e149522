"""Worker of tests/test_gpu_fortran.py-style multi-process test: run under torch.distributed.run
with 2 ranks sharing one GPU over gloo.  Each rank owns one slab of a 4 x 3 x 6 domain, steps the
preweight and the transport with CellBlock.transport_preweight_distributed /
.transport_distributed, and checks its real rows against pmcx_transport_preweight and
pmcx_transport on the whole domain (computed locally by every rank)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("PMCX_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import wrf_partmc_amd as wp
    from tests import parity
    nx, nz, ny, n_class, n_spec, cap, n_part = 4, 3, 3 * world, 2, 2, 1024, 400
    row, n = nx * nz, nx * nz * ny
    rng = np.random.RandomState(5)
    names, dens = parity.default_species(n_spec)
    modes, _ = parity.make_modes(n_spec, n_class)
    for m in modes:
        m["source"] = m["source"] % 2
    mag = rng.uniform(0.5e7, 2e7, (n, n_class)); cv = rng.uniform(0.8e9, 1.3e9, n)
    ph = rng.uniform(0, 0.08, (n, n_class, 4)); pv = rng.uniform(0, 0.04, (n, n_class, nz))
    whole = wp.CellBlock(n, cap, n_spec, n_class=n_class, n_source=2)
    whole.set_aero_data(dens); whole.set_env(280.0, 9e4); whole.set_weights(mag)
    whole.set_next_id((np.arange(n, dtype=np.int64) << 32) + (1 << 20))
    n_ideal = np.full((n, n_class), 1e9)
    n_ideal[[2, row + 1, 3 * row - 2, 3 * row + 5, n - 3]] = 60.0      # cells whose weights get scaled: both
    whole.set_n_part_ideal(n_ideal)                                     # sides of the cut and of the edge
    whole.generate_lognormal(n_part, modes, seed=3)
    dn = whole.download_cells()
    j0 = 3 * rank
    rows = [(j0 - 1) % ny, j0, j0 + 1, j0 + 2, (j0 + 3) % ny]
    cells = np.concatenate([np.arange(j * row, (j + 1) * row) for j in rows])
    real = cells[row:-row]
    b = wp.CellBlock(len(cells), cap, n_spec, n_class=n_class, n_source=2,
                     cell_id_base=((j0 - 1) * row) % (1 << 32))
    b.set_aero_data(dens); b.set_env(280.0, 9e4); b.set_weights(mag[cells])
    b.set_next_id((cells.astype(np.int64) << 32) + (1 << 20))
    b.set_n_part_ideal(n_ideal[cells])
    po = np.concatenate([[0], np.cumsum(np.diff(dn["part_offset"])[real])])
    sel = np.concatenate([np.arange(dn["part_offset"][c], dn["part_offset"][c + 1]) for c in real])
    b.upload_cells(row, po, dn["vol"][:, sel], dn["wclass"][sel], dn["id"][sel], dn["least_ct"][sel],
                   dn["greatest_ct"][sel], dn["n_orig"][:, sel])
    for g0 in (0, len(cells) - row):
        b.upload_cells(g0, np.zeros(row + 1, np.int64), np.zeros((n_spec, 0)))
    for step in range(3):
        ex_w, nr_w = whole.transport_preweight(nx, nz, ny, ph, pv, cv, 9, step, True, True)
        ex, nr = b.transport_preweight_distributed(dist, nx, nz, 5, ph[cells], pv[cells], cv[cells], ny, 9, step)
        assert np.array_equal(ex[row:-row], ex_w[real]), (rank, step, np.abs(ex[row:-row] - ex_w[real]).max())
        assert np.array_equal(nr[row:-row], nr_w[real]) and (step > 0 or nr_w.sum() > 0)
        assert np.array_equal(b.get_weights()[row:-row], whole.get_weights()[real])
        # (the ghost rows stand for the neighbours' rows: their weights followed the scaling there)
        assert np.array_equal(b.get_weights(), whole.get_weights()[cells])
        nm_w, na_w = whole.transport(nx, nz, ny, ph, pv, cv, 9, step, True, True)
        # the probabilities the preweight left on the device serve the transport
        nm, na = b.transport_distributed(dist, nx, nz, 5, None, None, cv[cells], 9, step)
        assert np.array_equal(nm[row:-row], nm_w[real]) and np.array_equal(na[row:-row], na_w[real])
        ref = whole.download_cells()
        got = b.download_cells(row, len(real))
        sel = np.concatenate([np.arange(ref["part_offset"][c], ref["part_offset"][c + 1]) for c in real])
        for k in ("vol", "wclass", "id", "least_ct", "greatest_ct", "n_orig"):
            assert np.array_equal(got[k], ref[k][..., sel]), (rank, step, k)
    # ---- the probabilities built on the device from WRF-shaped fields, slab by slab ----
    from tests import trans_fields as tf
    F = tf.make_fields(nx, nz, ny, n_class, seed=7, k_scale=5.0, flux_scale=0.03 * n_part * 1e7)
    F["v_flux"][:, ny] = F["v_flux"][:, 0]                               # one face: the domain is periodic in y
    rows5 = [(j0 - 1) % ny, j0, j0 + 1, j0 + 2, (j0 + 3) % ny]
    halo = lambda a: np.concatenate([a[-1:], a, a[:1]], axis=0)          # periodic rows around the domain
    alt_p, xk_p = halo(F["alt"][1:-1]), halo(F["xkhh"][1:-1])            # [ny+2][nz][nx+2], rows wrapped
    Fw = dict(F, alt=alt_p, xkhh=xk_p)
    # the slab's fields: its five rows (two ghosts) with their own halo rows
    rr = [(r - 1) % ny for r in rows5[:1]] + rows5 + [(rows5[-1] + 1) % ny]
    vf = F["v_flux"]                                                     # ny + 1 faces: face j lies below row j
    Fs = dict(alt=alt_p[[r + 1 for r in rr]], xkhh=xk_p[[r + 1 for r in rr]],
              u_flux=F["u_flux"][:, rows5], w_flux=F["w_flux"][:, rows5],
              v_flux=np.stack([vf[:, r] for r in rows5] + [vf[:, rows5[-1] + 1]], axis=1),
              z=F["z"][rows5], z_at_w=F["z_at_w"][rows5], exch_h=F["exch_h"][rows5])
    for step in range(3, 5):
        phw, pvw, nsw, badw, _ = whole.transport_probs(nx, nz, ny, 30.0, 3000.0, 3000.0, Fw)
        phs, pvs, nss, bads, _ = b.transport_probs(nx, nz, 5, 30.0, 3000.0, 3000.0, Fs, j_offset=j0 - 1,
                                                   global_nx=nx, global_ny=ny)
        assert not badw.any() and not bads[row:-row].any()
        assert np.array_equal(phs[row:-row], phw[real]) and np.array_equal(pvs[row:-row], pvw[real]), (rank, step)
        ex_w, nr_w = whole.transport_preweight(nx, nz, ny, None, None, cv, 9, step, True, True)
        ex, nr = b.transport_preweight_distributed(dist, nx, nz, 5, None, None, cv[cells], ny, 9, step)
        assert np.array_equal(ex[row:-row], ex_w[real]) and np.array_equal(nr[row:-row], nr_w[real]), (rank, step)
        nm_w, na_w = whole.transport(nx, nz, ny, None, None, cv, 9, step, True, True)
        nm, na = b.transport_distributed(dist, nx, nz, 5, None, None, cv[cells], 9, step)
        assert np.array_equal(nm[row:-row], nm_w[real]) and np.array_equal(na[row:-row], na_w[real]), (rank, step)
        ref = whole.download_cells()
        got = b.download_cells(row, len(real))
        sel = np.concatenate([np.arange(ref["part_offset"][c], ref["part_offset"][c + 1]) for c in real])
        for k in ("vol", "wclass", "id", "least_ct", "greatest_ct", "n_orig"):
            assert np.array_equal(got[k], ref[k][..., sel]), (rank, step, k)
    whole.close(); b.close()

    # ---- open lateral boundaries (neither direction periodic), the refill included ----
    whole = wp.CellBlock(n, cap, n_spec, n_class=n_class, n_source=2)
    whole.set_aero_data(dens); whole.set_env(280.0, 9e4); whole.set_weights(mag)
    whole.set_next_id((np.arange(n, dtype=np.int64) << 32) + (1 << 20))
    whole.set_n_part_ideal(np.full((n, n_class), 1e9))
    whole.generate_lognormal(n_part, modes, seed=4)
    whole.set_transport_lateral(True)
    dn = whole.download_cells()
    rows = [j0 - 1, j0, j0 + 1, j0 + 2, j0 + 3]                    # (rows outside the domain: unused ghosts)
    cells = np.concatenate([np.arange((j % ny) * row, ((j % ny) + 1) * row) for j in rows])
    real = cells[row:-row]
    b = wp.CellBlock(len(cells), cap, n_spec, n_class=n_class, n_source=2,
                     cell_id_base=((j0 - 1) * row) % (1 << 32))
    b.set_aero_data(dens); b.set_env(280.0, 9e4); b.set_weights(mag[cells])
    b.set_next_id((cells.astype(np.int64) << 32) + (1 << 20))
    b.set_n_part_ideal(np.full((len(cells), n_class), 1e9))
    b.set_transport_lateral(True, j0 - 1, ny)
    po = np.concatenate([[0], np.cumsum(np.diff(dn["part_offset"])[real])])
    sel = np.concatenate([np.arange(dn["part_offset"][c], dn["part_offset"][c + 1]) for c in real])
    b.upload_cells(row, po, dn["vol"][:, sel], dn["wclass"][sel], dn["id"][sel], dn["least_ct"][sel],
                   dn["greatest_ct"][sel], dn["n_orig"][:, sel])
    for g0 in (0, len(cells) - row):
        b.upload_cells(g0, np.zeros(row + 1, np.int64), np.zeros((n_spec, 0)))
    bg = [dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[0.5, 0.5], wclass=0, source=0),
          dict(gm_radius=1e-7, gsd_log10=0.2, vol_frac=[1.0, 0.0], wclass=1, source=1)]
    rho = rng.uniform(0.9, 1.2, n)
    bg_conc = np.stack([0.8 * n_part / n_class * mag[:, 0] / rho, 1.3 * n_part / n_class * mag[:, 1] / rho], axis=1)
    for step in range(2):
        u_w = rng.normal(0, 1, (ny, nz)); u_e = rng.normal(0, 1, (ny, nz))
        v_s = rng.normal(0, 1, (nx, nz)); v_n = rng.normal(0, 1, (nx, nz))
        nm_w, na_w = whole.transport(nx, nz, ny, ph, pv, cv, 21, step, False, False)
        out_w = whole.lateral_boundary_conditions(nx, nz, ny, u_w, u_e, v_s, v_n, bg, bg_conc, rho, seed=21, step=step)
        nm, na = b.transport_distributed(dist, nx, nz, 5, ph[cells], pv[cells], cv[cells], 21, step, False, False)
        assert np.array_equal(nm[row:-row], nm_w[real]) and np.array_equal(na[row:-row], na_w[real]), (rank, step)
        lj = [(j % ny) for j in rows]
        out = b.lateral_boundary_conditions(nx, nz, 5, u_w[lj], u_e[lj], v_s, v_n, bg, bg_conc[cells], rho[cells],
                                            seed=21, step=step, j_global0=j0 - 1, global_ny=ny)
        assert len(out["copied"]) + len(out["resampled"]) > 0
        ref = whole.download_cells()
        got = b.download_cells(row, len(real))
        sel = np.concatenate([np.arange(ref["part_offset"][c], ref["part_offset"][c + 1]) for c in real])
        assert np.array_equal(np.diff(got["part_offset"]), np.diff(ref["part_offset"])[real]), (rank, step)
        for k in ("vol", "wclass", "id", "least_ct", "greatest_ct", "n_orig"):
            assert np.array_equal(got[k], ref[k][..., sel]), (rank, step, k)
        assert np.array_equal(b.get_weights()[row:-row], whole.get_weights()[real])
    dist.barrier()
    if rank == 0:
        print(f"distributed transport ok: {world} ranks, {int(nm_w.sum())} movers in the last step")
    whole.close(); b.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
