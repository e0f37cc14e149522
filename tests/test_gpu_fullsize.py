"""-m gpu: BASELINE configs[4] at its stated size (100 x 100 x 30 cells x 10^4 particles) and at the
shares of it a rank holds at N = 4 and 8, with the library's own choice of sampler
(launch_coag_sample: lane per cell above 65 536 cells, its two-wavefront form from 24 000).

What is checked (interface/wrf_pmc_driver.F90:169-198, the loop body over every cell):
  * on every cell: status 0, n_part after = n_part before - sum of n_coag (one weight class),
    mass concentration conserved to 1e-12 (tree-summed fp64);
  * on >= 256 cells spread over the whole id range -- the first and the last workgroup, cells
    beyond 213 730 (slot indices past 2^31 at cap 10 048) -- after 3 steps of mc_coag + rebalance:
    n_samp, n_coag, status of every step and every downloaded particle field equal to the CPU
    oracle's, bit for bit; ids unique.
Sized by the free HBM of the box: the full domain needs about 205 GB."""
import numpy as np
import pytest

import bench
from tests import parity  # noqa: F401  (keeps the import order of the other GPU modules)

pytestmark = pytest.mark.gpu

K_BROWN = 4
N_PART, CAP, LEVELS = 10000, 10048, 30
NUM_CONC, DEL_T, SEED, STEPS = 1e10, 300.0, 0, 3


def _sample_cells(n_cells, first_cell):
    """Local indices of the cells held against the oracle."""
    loc = set(np.linspace(0, n_cells - 1, 192).astype(int).tolist())
    loc.update(range(0, 8))                                  # first workgroup
    loc.update(range(n_cells - 64, n_cells))                 # last workgroup
    for g in (213730, 213731, 214000, 214001, 214748, 250001, 262143, 262144, 299999):
        if first_cell <= g < first_cell + n_cells:           # slot indices past 2^31, 2^18 cells
            loc.add(g - first_cell)
    return np.array(sorted(loc), dtype=np.int64)


def _run_block(wp, oracle, n_cells, first_cell):
    free, total = wp.device_memory_info(0)
    need = n_cells * (CAP * 70 + 80 * 80 * 4 + 4096) + (2 << 30)
    if free < need:
        pytest.skip(f"{n_cells} cells x {N_PART} need {need / 1e9:.0f} GB of HBM, {free / 1e9:.0f} free")
    blk = wp.CellBlock(n_cells, CAP, 1, n_class=1, n_source=0, cell_id_base=first_cell,
                       max_events=1024, max_appends=128)
    temp, pres = bench.level_profile(LEVELS)
    lev = (first_cell + np.arange(n_cells)) % LEVELS
    dens = bench.bench_density(1)
    blk.set_aero_data(dens)
    blk.set_env(temp[lev], pres[lev], np.full(n_cells, 0.5))
    blk.set_weights(np.full((n_cells, 1), NUM_CONC / N_PART))
    blk.set_n_part_ideal(np.full((n_cells, 1), float(N_PART)))
    blk.set_next_id((first_cell + np.arange(n_cells, dtype=np.int64) << 32) + (1 << 24))
    modes = bench.bench_modes(1)
    blk.generate_lognormal(N_PART, modes, seed=SEED)
    blk.update_kernel_bounds(K_BROWN)

    sample = _sample_cells(n_cells, first_cell)
    assert len(sample) >= 256
    ve, vs = blk.bin_tables()
    cells, kcache = {}, {}
    for k in sample:
        gid = first_cell + int(k)
        lv = gid % LEVELS
        oc = oracle.Cell(1, 1, 0, CAP, dens, ve, temp[lv], pres[lv], max_events=1024, max_appends=128)
        oc.mag[:] = NUM_CONC / N_PART
        oc.generate(modes, N_PART, SEED, gid, first_id=gid << 32)
        if lv not in kcache:
            kcache[lv] = oracle.kmax_table(K_BROWN, vs, dens.min(), dens.max(), temp[lv], pres[lv])
        oc.kmax[:] = kcache[lv]
        oc.next_id = int((gid << 32) + (1 << 24))
        cells[int(k)] = (gid, oc)

    n0 = blk.aero_state_total_particles().astype(np.int64)
    assert (n0 == N_PART).all()
    _, mass0, _, _ = blk.diagnostics()
    tot_coag = np.zeros(n_cells, np.int64)
    for step in range(STEPS):
        ns, nc, st = blk.mc_coag(K_BROWN, DEL_T, SEED, step)
        assert not st.any(), f"step {step}: status bits {np.unique(st)} on {np.count_nonzero(st)} cells"
        tot_coag += nc
        blk.aero_state_rebalance(True, True, SEED, step)
        for k, (gid, oc) in cells.items():
            ost, ons, onc = oc.mc_coag(K_BROWN, DEL_T, SEED, gid, step)
            assert (ons, onc, ost) == (int(ns[k]), int(nc[k]), int(st[k])), \
                f"step {step} cell {gid}: gpu {(int(ns[k]), int(nc[k]), int(st[k]))} oracle {(ons, onc, ost)}"
            _, oc.next_id = oc.rebalance(np.full(1, float(N_PART)), True, True, SEED, gid, step, oc.next_id)
    assert tot_coag.min() > 0
    n1 = blk.aero_state_total_particles().astype(np.int64)
    assert np.array_equal(n1, n0 - tot_coag)
    _, mass1, _, _ = blk.diagnostics()
    np.testing.assert_allclose(mass1, mass0, rtol=1e-12)

    for k, (gid, oc) in cells.items():
        dn = blk.download_cells(int(k), 1)
        n = int(oc.n_part[0])
        assert dn["part_offset"][1] == n, f"cell {gid}: n_part gpu {dn['part_offset'][1]} oracle {n}"
        for name, a, b in (("vol", dn["vol"][0], oc.vol[0, :n]), ("id", dn["id"], oc.id[:n]),
                           ("least_ct", dn["least_ct"], oc.lct[:n]),
                           ("greatest_ct", dn["greatest_ct"], oc.gct[:n])):
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8),
                                  np.ascontiguousarray(b).view(np.uint8)), f"cell {gid}: {name} differs"
        assert len(np.unique(dn["id"])) == n
    blk.close()
    return int(tot_coag.sum())


def test_full_domain_300000_cells(wp, oracle):
    """BASELINE configs[4] on one GPU: 300 000 cells, lane-per-cell sampler (native dispatch)."""
    ev = _run_block(wp, oracle, 100 * 100 * LEVELS, 0)
    assert ev > 3 * 10 ** 7


@pytest.mark.parametrize("n_cols,first_col", [(2500, 2500), (1300, 8700)])
def test_rank_share_of_the_domain(wp, oracle, n_cols, first_col):
    """What a rank holds at N = 4 (75 000 cells) and N = 8 (39 000 cells: the last rank's
    block, two-wavefront form of the sampler), keyed by global cell ids."""
    _run_block(wp, oracle, n_cols * LEVELS, first_col * LEVELS)


def test_transport_driver_on_the_full_domain(wp, oracle):
    """wrf_pmc_trans_driver's particle half at BASELINE configs[4]'s grid (100 x 30 x 100 cells in
    WRF's (i, k, j) order, 9 000 particles each so that arrivals find room below cap): the
    probabilities built on the device equal the oracle's for EVERY cell and class bit for bit, the
    preweight's expected concentrations equal an array-wise numpy evaluation, and the sampling
    that consumes both from device memory conserves particles cell by cell -- with slot and
    outbox indices past 2^31 in play (cells beyond 213 730)."""
    from tests import trans_fields as tf
    nx, nz, ny, n_part = 100, LEVELS, 100, 9000
    n_cells = nx * nz * ny
    free, total = wp.device_memory_info(0)
    need = n_cells * (CAP * 78 + 80 * 80 * 4 + 4096 + 8 * (4 + nz) * 2) + (3 << 30)
    if free < need:
        pytest.skip(f"the transport at {n_cells} cells needs {need / 1e9:.0f} GB of HBM, {free / 1e9:.0f} free")
    blk = wp.CellBlock(n_cells, CAP, 1, n_class=1, n_source=0, max_events=1024, max_appends=128)
    temp, pres = bench.level_profile(LEVELS)
    k = (np.arange(n_cells) // nx) % nz
    blk.set_aero_data(bench.bench_density(1))
    blk.set_env(temp[k], pres[k], np.full(n_cells, 0.5))
    mag = NUM_CONC / n_part
    blk.set_weights(np.full((n_cells, 1), mag))
    blk.set_n_part_ideal(np.full((n_cells, 1), float(n_part)))
    blk.set_next_id((np.arange(n_cells, dtype=np.int64) << 32) + (1 << 24))
    blk.generate_lognormal(n_part, bench.bench_modes(1), seed=SEED)
    # (exchange coefficients of a few m^2/s: with 40 m levels and a 20 s step a few per cent of a
    # cell's particles change level; stronger mixing would need more room than cap leaves)
    F = tf.make_fields(nx, nz, ny, 1, seed=11, k_scale=6.0, flux_scale=0.004 * NUM_CONC)
    kw = dict(do_vertmix=1, all_columns=True)
    ph, pv, ns, bad, ncp = blk.transport_probs(nx, nz, ny, 20.0, 4000.0, 4000.0, F, **kw)
    assert not bad.any() and (ncp == n_part).all() and ns.min() >= 1 and ns.max() > 1
    oph, opv, ons, obad = oracle.transport_probs(nx, nz, ny, 1, 20.0, 4000.0, 4000.0, F,
                                                 np.full((n_cells, 1), n_part), np.full((n_cells, 1), mag), **kw)
    assert np.array_equal(ns, ons) and np.array_equal(ph, oph) and np.array_equal(pv, opv)
    cv = np.full(n_cells, 1.0e9)
    ex, nr = blk.transport_preweight(nx, nz, ny, None, None, cv, SEED, 0, True, True)
    # numpy: losses and gains per direction with equal cell volumes (sums in another order)
    O = np.full((ny, nz, nx), n_part * mag)
    PH, PV = ph.reshape(ny, nz, nx, 4), pv.reshape(ny, nz, nx, nz)
    new = O.copy()
    for d, (ax, sh) in enumerate([(2, -1), (2, 1), (0, -1), (0, 1)]):
        flux = PH[..., d] * O
        new -= flux
        new += np.roll(flux, sh, axis=ax)
    own = np.eye(nz, dtype=bool)[None, :, None, :]
    fl = np.where(own, 0.0, PV * O[..., None])                    # [j][k_src][i][k_dst]
    new -= fl.sum(axis=3)
    new += np.moveaxis(fl.sum(axis=1), 2, 1)
    np.testing.assert_allclose(ex[:, 0], new.reshape(-1), rtol=1e-12)
    assert not nr.any()                                           # nowhere near twice the ideal count
    n0 = blk.aero_state_total_particles().astype(np.int64)
    nm, na = blk.transport(nx, nz, ny, None, None, cv, SEED, 0, True, True)
    n1 = blk.aero_state_total_particles().astype(np.int64)
    assert not blk.coag_counters()[2].any(), "status bits after the transport (capacity?)"
    assert np.array_equal(n1, n0 - nm + na) and nm.sum() == na.sum() and np.count_nonzero(nm) > 0.99 * n_cells
    assert 0.005 * n_part < nm.mean() < 0.2 * n_part
    # the movers of a cell follow its probabilities (a column of cells past 2^31 in slot index)
    p_leave = ph[:, 0].sum(axis=1) + pv[:, 0].sum(axis=1) - pv[np.arange(n_cells), 0, k]
    far = np.arange(n_cells - 3000, n_cells)
    assert abs(nm[far].sum() - (p_leave[far] * n_part).sum()) < 5 * np.sqrt((p_leave[far] * n_part).sum())
    for c in (0, 213731, 250001, n_cells - 1):
        dn = blk.download_cells(c, 1)
        assert dn["part_offset"][1] == n1[c] and len(np.unique(dn["id"])) == n1[c]
        assert (dn["vol"][0] > 0).all()
    # and the coagulation step copes with what the transport left (lists rebuilt from the keys)
    ns_, nc_, st = blk.mc_coag(K_BROWN, DEL_T, SEED, 0)
    assert not st.any() and nc_.min() > 0
    blk.close()


def test_bench_starts_its_own_ranks_on_the_gpu():
    """`python bench.py --gpus 2` with no launcher (no WORLD_SIZE in the environment): bench.py starts
    its two ranks itself -- children, before it touches the GPU -- and passes on rank 0's ONE line.
    Two ranks share this box's GPU over gloo; the events of the two column blocks must be those of
    the same domain on one rank (RNG keyed by the global cell id)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--nx", "6", "--ny", "4", "--levels", "5", "--particles", "2000",
              "--no-cpu-baseline", "--resort-steps", "0"]
    lines = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--backend", "gloo"]
                           + common, env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1, r.stdout
        lines[n] = json.loads(js[0])
    assert lines[2]["n_gpus"] == 2 and lines[1]["n_gpus"] == 1
    assert lines[2]["events_per_step"] == lines[1]["events_per_step"] > 0
    assert lines[2]["status_or"] == 0
