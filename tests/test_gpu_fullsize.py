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
