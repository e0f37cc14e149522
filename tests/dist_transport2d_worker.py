"""Worker of tests/test_gpu_fortran.py::test_transport_across_a_grid_of_ranks: run under
torch.distributed.run with 4 ranks sharing one GPU over gloo, a 2 x 2 grid of blocks over an
8 x 3 x 6 domain cut in x AND y.  Every rank steps the preweight and the transport of its block with
CellBlock.transport_preweight_distributed_2d / .transport_distributed_2d and checks its real cells
against pmcx_transport_preweight and pmcx_transport on the whole domain (computed locally by every
rank), doubly periodic and with open lateral boundaries."""
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
    assert world == 4
    torch.cuda.set_device(0)
    import wrf_partmc_amd as wp
    from tests import parity
    grid = (2, 2)
    nx, nz, ny, n_class, n_spec, cap, n_part = 8, 3, 6, 2, 2, 1024, 400
    rx, ry = nx // grid[0], ny // grid[1]
    a, b = rank % grid[0], rank // grid[0]
    n = nx * nz * ny
    names, dens = parity.default_species(n_spec)
    modes, _ = parity.make_modes(n_spec, n_class)
    for m in modes:
        m["source"] = m["source"] % 2
    for periodic in (True, False):
        rng = np.random.RandomState(23)
        mag = rng.uniform(0.5e7, 2e7, (n, n_class)); cv = rng.uniform(0.8e9, 1.3e9, n)
        ph = rng.uniform(0, 0.08, (n, n_class, 4)); pv = rng.uniform(0, 0.04, (n, n_class, nz))
        whole = wp.CellBlock(n, cap, n_spec, n_class=n_class, n_source=2)
        whole.set_aero_data(dens); whole.set_env(280.0, 9e4); whole.set_weights(mag)
        whole.set_next_id((np.arange(n, dtype=np.int64) << 32) + (1 << 20))
        whole.set_n_part_ideal(np.full((n, n_class), 0.45 * n_part / n_class))
        whole.generate_lognormal(n_part, modes, seed=3)
        if not periodic:
            whole.set_transport_lateral(True)
        dn = whole.download_cells()
        gi0, gj0, nxl, nyl = a * rx - 1, b * ry - 1, rx + 2, ry + 2
        lj, lk, li = np.meshgrid(np.arange(nyl), np.arange(nz), np.arange(nxl), indexing="ij")
        gmap = ((((gj0 + lj) % ny) * nz + lk) * nx + (gi0 + li) % nx).ravel()
        real = ((li >= 1) & (li <= rx) & (lj >= 1) & (lj <= ry)).ravel()
        gl = np.flatnonzero(real)
        blk = wp.CellBlock(len(gmap), cap, n_spec, n_class=n_class, n_source=2,
                           cell_id_base=((gj0 * nz) * nx + gi0) % (1 << 32))
        blk.set_transport_cuts(True, nxl, gi0, nx)
        if not periodic:
            blk.set_transport_lateral(True, gj0, ny)
        blk.set_aero_data(dens); blk.set_env(280.0, 9e4); blk.set_weights(mag[gmap])
        blk.set_next_id((gmap.astype(np.int64) << 32) + (1 << 20))
        blk.set_n_part_ideal(np.full((len(gmap), n_class), 0.45 * n_part / n_class))
        counts = np.where(real, np.diff(dn["part_offset"])[gmap], 0)
        po = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        sel = np.concatenate([np.arange(dn["part_offset"][g], dn["part_offset"][g + 1]) for g in gmap[gl]])
        blk.upload_cells(0, po, dn["vol"][:, sel], dn["wclass"][sel], dn["id"][sel], dn["least_ct"][sel],
                         dn["greatest_ct"][sel], dn["n_orig"][:, sel])
        moved = 0
        for step in range(3):
            ex_w, nr_w = whole.transport_preweight(nx, nz, ny, ph, pv, cv, 9, step, periodic, periodic)
            ex, nr = blk.transport_preweight_distributed_2d(dist, grid, nxl, nz, nyl, ph[gmap], pv[gmap], cv[gmap],
                                                            gj0, ny, 9, step, periodic, periodic)
            assert np.array_equal(ex[real], ex_w[gmap[real]]), (rank, step, periodic)
            assert np.array_equal(nr[real], nr_w[gmap[real]]) and (step > 0 or nr_w.sum() > 0)
            w_now = whole.get_weights()
            assert np.array_equal(blk.get_weights()[real], w_now[gmap[real]])
            # every ghost cell that stands for a cell of the domain (no corner, nothing beyond an open
            # boundary) has its owner's new weight
            has = wp.host.grid_neighbours(rank, grid, periodic, periodic)
            ghost_ok = ((lj == 0) & (li >= 1) & (li <= rx) & (has[0] is not None)) | \
                       ((lj == nyl - 1) & (li >= 1) & (li <= rx) & (has[1] is not None)) | \
                       ((li == 0) & (lj >= 1) & (lj <= ry) & (has[2] is not None)) | \
                       ((li == nxl - 1) & (lj >= 1) & (lj <= ry) & (has[3] is not None))
            ghost_ok = ghost_ok.ravel()
            assert np.array_equal(blk.get_weights()[ghost_ok], w_now[gmap[ghost_ok]]), (rank, step, periodic)
            nm_w, na_w = whole.transport(nx, nz, ny, ph, pv, cv, 9, step, periodic, periodic)
            nm, na = blk.transport_distributed_2d(dist, grid, nxl, nz, nyl, None, None, cv[gmap], 9, step,
                                                  periodic, periodic)
            assert np.array_equal(nm[real], nm_w[gmap[real]]) and np.array_equal(na[real], na_w[gmap[real]])
            assert not blk.aero_state_total_particles()[~real].any()
            if not periodic:
                # the refill of the lateral boundary cells (wrf_pmc_trans_aero.F90:803-955): every rank
                # sees to the boundary cells it holds
                r2 = np.random.RandomState(40 + step)
                u_w = r2.normal(0, 1, (ny, nz)); u_e = r2.normal(0, 1, (ny, nz))
                v_s = r2.normal(0, 1, (nx, nz)); v_n = r2.normal(0, 1, (nx, nz))
                rho = r2.uniform(0.9, 1.2, n)
                bg = [dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[0.5, 0.5], wclass=0, source=0),
                      dict(gm_radius=1e-7, gsd_log10=0.2, vol_frac=[1.0, 0.0], wclass=1, source=1)]
                bg_conc = np.stack([0.8 * n_part / n_class * mag[:, 0] / rho, 1.3 * n_part / n_class * mag[:, 1] / rho], axis=1)
                whole.lateral_boundary_conditions(nx, nz, ny, u_w, u_e, v_s, v_n, bg, bg_conc, rho, seed=21, step=step)
                gj = (gj0 + np.arange(nyl)) % ny; gi = (gi0 + np.arange(nxl)) % nx
                out = blk.lateral_boundary_conditions(nxl, nz, nyl, u_w[gj], u_e[gj], v_s[gi], v_n[gi], bg, bg_conc[gmap],
                                                      rho[gmap], seed=21, step=step, j_global0=gj0, global_ny=ny,
                                                      i_global0=gi0, global_nx=nx)
                assert len(out["copied"]) + len(out["resampled"]) > 0
            ref, got = whole.download_cells(), blk.download_cells()
            assert np.array_equal(np.diff(got["part_offset"])[gl], np.diff(ref["part_offset"])[gmap[gl]])
            sel = np.concatenate([np.arange(ref["part_offset"][g], ref["part_offset"][g + 1]) for g in gmap[gl]])
            for k in ("vol", "wclass", "id", "least_ct", "greatest_ct", "n_orig"):
                assert np.array_equal(got[k], ref[k][..., sel]), (rank, step, periodic, k)
            moved += int(nm[real].sum())
        assert moved > 100
        if periodic:
            # the probabilities built on the device from WRF-shaped fields, block by block
            from tests import trans_fields as tf
            F = tf.make_fields(nx, nz, ny, n_class, seed=7, k_scale=5.0, flux_scale=0.03 * n_part * 1e7)
            F["v_flux"][:, ny] = F["v_flux"][:, 0]; F["u_flux"][..., nx] = F["u_flux"][..., 0]   # one face each: periodic
            wrap = lambda c: np.pad(c, ((1, 1), (0, 0), (1, 1)), mode="wrap")
            Fw = dict(F, alt=wrap(F["alt"][1:-1, :, 1:-1]), xkhh=wrap(F["xkhh"][1:-1, :, 1:-1]))
            rows, cols = (gj0 + np.arange(nyl)) % ny, (gi0 + np.arange(nxl)) % nx
            rows_h, cols_h = (gj0 - 1 + np.arange(nyl + 2)) % ny, (gi0 - 1 + np.arange(nxl + 2)) % nx
            fx, fy = (gi0 + np.arange(nxl + 1)) % nx, (gj0 + np.arange(nyl + 1)) % ny
            core = lambda k: Fw[k][1:-1, :, 1:-1]
            Fs = dict(alt=core("alt")[rows_h][:, :, cols_h], xkhh=core("xkhh")[rows_h][:, :, cols_h],
                      u_flux=F["u_flux"][:, rows][..., fx], v_flux=F["v_flux"][:, fy][..., cols],
                      w_flux=F["w_flux"][:, rows][..., cols], z=F["z"][rows][..., cols],
                      z_at_w=F["z_at_w"][rows][..., cols], exch_h=F["exch_h"][rows][..., cols])
            phw, pvw, nsw, badw, _ = whole.transport_probs(nx, nz, ny, 30.0, 3000.0, 3000.0, Fw)
            phs, pvs, nss, bads, _ = blk.transport_probs(nxl, nz, nyl, 30.0, 3000.0, 3000.0, Fs, i_offset=gi0,
                                                         j_offset=gj0, global_nx=nx, global_ny=ny)
            assert not badw.any() and not bads[real].any()
            assert np.array_equal(phs[real], phw[gmap[real]]) and np.array_equal(pvs[real], pvw[gmap[real]]), rank
        whole.close(); blk.close()
    dist.barrier()
    if rank == 0:
        print("dist_transport2d_worker ok: 4 ranks, 2 x 2 blocks, periodic and open boundaries")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
