"""-m gpu: growing particle arrays.  PartMC's aero_particle_array grows when a particle is added to a
full one (behind aero_state_add_particle, interface/wrf_pmc_trans_aero.F90:777-804, 2284); a block here
has cap slots per cell.  pmcx_grow re-pitches the block; pmcx_set_auto_grow lets the library do it --
exactly, before a transport gathers arrivals a cell would not hold; by a margin at the start of
pmcx_timestep and of the boundary refill.  No result depends on cap: every test holds a block that
started small to the oracle, whose arrays are large from the start."""
import numpy as np
import pytest

from tests import parity
from tests.test_gpu_parity import _transport_probs

pytestmark = pytest.mark.gpu

K_BROWN = 4


def _small_twin(wp, blk, cap, **over):
    """A block with the particles, weights, ideal counts and ids of `blk` in cells of `cap` slots."""
    cf = blk.cfg
    kw = dict(n_class=cf.n_class, n_source=cf.n_source, cell_id_base=cf.cell_id_base, max_events=cf.max_events,
              max_appends=cf.max_appends, n_comp_max=cf.n_comp_max)
    kw.update(over)
    tw = wp.CellBlock(cf.n_cells, cap, cf.n_spec, **kw)
    tw.set_aero_data(blk._density)
    dn = blk.download_cells()
    tw.upload_cells(0, dn["part_offset"], dn["vol"], dn["wclass"], dn["id"], dn["least_ct"], dn["greatest_ct"],
                    dn["n_orig"] if cf.n_source > 0 else None)
    if cf.n_comp_max > 0:
        cm = blk.download_components()
        tw.upload_components(0, dn["part_offset"], cm["comp_len"], cm["comp_source"], cm["comp_time"])
    tw.set_weights(blk.get_weights())
    tw.set_next_id(blk.get_next_id())
    return tw


def test_grow_keeps_every_particle_list_and_key(wp, oracle):
    """pmcx_grow between steps: cells of 1024 slots that hold lists kept from the last mc_coag, tombstones,
    an event log and component lists are re-pitched to 1536 and then 4096 slots; mc_coag (its kept lists:
    no sort runs), a transport, a rebalance and the diagnostics go on as if nothing had happened --
    against the oracle at every step."""
    nx, nz, ny = 2, 3, 2
    n_cells, n_part, n_spec, n_class = nx * nz * ny, 900, 2, 2
    blk, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, 41, K_BROWN, cap=1024, n_source=2,
                                         num_conc_scale=10.0, n_comp_max=4)
    ph, pv, cv = _transport_probs(n_cells, n_class, nz, 3)
    next_id = np.array([oc.next_id for oc in cells], dtype=np.int64)
    ideal = np.full(n_class, n_part / n_class)
    assert blk.growth() == (1024, 0)
    for step in range(4):
        ns, nc, st = blk.mc_coag(K_BROWN, 900.0, 41, step)
        for c, oc in enumerate(cells):
            assert oc.mc_coag(K_BROWN, 900.0, 41, oc.cell_id, step) == (int(st[c]), int(ns[c]), int(nc[c]))
        if step == 0:
            assert blk.grow(1500) == 1536 and blk.growth() == (1536, 1)       # (rounded up to 64)
            perm, gcnt, goff = blk.get_sort(0)                                # the lists moved with the block
            assert gcnt.sum() == int(cells[0].n_part[0])
        if step == 2:
            assert blk.grow(4096) == 4096 and blk.grow(100) == 4096           # (never shrinks)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, f"step {step} after mc_coag: {msg}"
        nm, na = blk.transport(nx, nz, ny, ph, pv, cv, 41, step, True, True)
        sto, onm, ona = oracle.transport(cells, nx, nz, ny, ph, pv, cv, 41, 0, step, next_id, True, True)
        assert np.array_equal(nm, onm) and np.array_equal(na, ona) and sto == 0
        blk.aero_state_rebalance(True, True, 41, step)
        for c, oc in enumerate(cells):
            _, next_id[c] = oc.rebalance(ideal, True, True, 41, oc.cell_id, step, int(next_id[c]))
        ok, msg = parity.compare_state(blk, cells)
        assert ok, f"step {step} after transport and rebalance: {msg}"
        assert np.array_equal(blk.get_next_id(), next_id)
    with pytest.raises(wp.PmcxError):
        blk.grow(70000)
    blk.close()


@pytest.mark.parametrize("law", [1, 0])
def test_a_transport_that_would_overflow_grows_the_block_first(wp, oracle, law):
    """The transient the Fortran shim's headroom was for: cells whose weight is a fraction of their
    neighbours' take 1/ratio copies of every arrival.  The oracle runs with room to spare; its twin on the
    device starts with cells that hold their particles and little more -- without pmcx_set_auto_grow the
    transport raises PMCX_ST_CAP_OVERFLOW and drops arrivals, with it the block grows before anything is
    gathered (the arrivals are counted by their sources first) and equals the oracle bit for bit."""
    nx, nz, ny = 3, 4, 2
    n_cells, n_part, n_spec, n_class = nx * nz * ny, 800, 1, 1
    big, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, 43, K_BROWN, cap=32768, n_source=0)
    mag = info["mag"] * 10 ** np.random.RandomState(2).uniform(-0.9, 0.9, (n_cells, 1))
    big.set_weights(mag)
    for c, oc in enumerate(cells):
        oc.mag[:] = mag[c]
    ph, pv, cv = _transport_probs(n_cells, n_class, nz, 11, scale=0.12)
    next_id = np.array([oc.next_id for oc in cells], dtype=np.int64)
    oracle.set_transport_law(law)
    try:
        small = _small_twin(wp, big, 832)
        small.set_transport_law(law)
        nm, na = small.transport(nx, nz, ny, ph, pv, cv, 43, 0, True, True)
        assert int(np.bitwise_or.reduce(small.coag_counters()[2])) & 16            # a full cell, arrivals lost
        small.close()
        small = _small_twin(wp, big, 832)
        small.set_transport_law(law)
        small.set_auto_grow(True)
        for step in range(3):
            nm, na = small.transport(nx, nz, ny, ph, pv, cv, 43, step, True, True)
            st, onm, ona = oracle.transport(cells, nx, nz, ny, ph, pv, cv, 43, 0, step, next_id, True, True)
            assert st == 0 and not np.bitwise_or.reduce(small.coag_counters()[2])
            assert np.array_equal(nm, onm) and np.array_equal(na, ona), step
            ok, msg = parity.compare_state(small, cells)
            assert ok, f"step {step}: {msg}"
            assert np.array_equal(small.get_next_id(), next_id)
        cap, n_grown = small.growth()
        assert n_grown >= 1 and cap >= max(int(oc.n_part[0]) for oc in cells) and cap < 32768
    finally:
        oracle.set_transport_law()
    small.close(); big.close()


def test_a_step_that_fills_cells_grows_the_block(wp, oracle):
    """pmcx_timestep with emissions into cells near their last slot: with pmcx_set_auto_grow the fullest
    cell is looked at when the step starts and the block grows by its margin; five steps of emission,
    dilution, background, coagulation and rebalance against the oracle, no status bit.  (The margin is
    not a count: an emission larger than a quarter of a cell in one step would still overflow.)"""
    n_cells, n_part, n_spec, n_class = 6, 700, 2, 2
    big, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, 47, K_BROWN, cap=16384, n_source=3)
    small = _small_twin(wp, big, 768)
    small.set_env(info["temp"], info["pres"], np.full(n_cells, 0.5))
    small.set_auto_grow(True, 0.75, 1.5)
    ideal = np.full((n_cells, n_class), 1.9 * n_part / n_class)      # (neither doubling nor halving: the cells fill up)
    small.set_n_part_ideal(ideal)
    mag = info["mag"]
    em = [dict(gm_radius=2e-8, gsd_log10=0.2, vol_frac=[0.5, 0.5], wclass=0, source=1),
          dict(gm_radius=1e-7, gsd_log10=0.1, vol_frac=[0.0, 1.0], wclass=1, source=2)]
    em_rate = np.array([[mag[c, 0] * 2.0, mag[c, 1] * 1.0] for c in range(n_cells)])
    dil = np.zeros(n_cells)
    for k in range(5):
        t = 3600.0 + 60.0 * k
        small.partmc_timestep(t, 60.0, 60.0, do_coagulation=True, kernel_type=K_BROWN, random_seed=47,
                              emission=em, emission_rate=em_rate, dilution_rate=dil)
        ns, nc, st = small.coag_counters()
        assert not np.bitwise_or.reduce(st), k
        for c, oc in enumerate(cells):
            s2, ons, onc = oc.mc_coag(K_BROWN, 60.0, 47, oc.cell_id, int(t / 60.0))
            assert (s2, ons, onc) == (0, int(ns[c]), int(nc[c])), (k, c)
            oe = [dict(m, num_conc=em_rate[c, j]) for j, m in enumerate(em)]
            s3, oc.next_id, _, _, _ = oc.scenario_update(60.0, t, 47, oc.cell_id, int(t / 60.0), oe, 0.0, [], oc.next_id)
            assert s3 == 0
            _, oc.next_id = oc.rebalance(ideal[c], True, True, 47, oc.cell_id, int(t / 60.0), oc.next_id)
        ok, msg = parity.compare_state(small, cells)
        assert ok, f"step {k}: {msg}"
        assert np.array_equal(small.get_next_id(), [oc.next_id for oc in cells]), k
    cap, n_grown = small.growth()
    assert n_grown >= 1 and cap > 768
    small.close(); big.close()
