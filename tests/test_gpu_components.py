"""-m gpu: particle%component(:) on the device (VERDICT a5; interface/wrf_pmc_driver.F90:1421-1437).
A context created with n_comp_max > 0 carries, per particle, the source_id and create_time of its
first n_comp_max constituents through every entry point that creates, copies, moves or merges
particles.  The parity tests of test_gpu_parity.py / test_gpu_timestep.py that change particles are
run again here with the lists switched on in both the block and the oracle cells
(parity.N_COMP_MAX): parity.compare_state then holds source ids and create times to the oracle bit
for bit, next to every other field, after mc_coag (all sampler forms), rebalance, the scenario
update, sea salt, transport (whole domain, slabs across two blocks, open lateral boundaries),
deposition and compaction."""
import numpy as np
import pytest

from tests import parity
from tests.test_gpu_parity import sampler_variant  # noqa: F401  (autouse: every sampler form)
from tests.test_gpu_parity import (  # noqa: F401
    test_mc_coag_parity, test_mc_coag_dense_events, test_rebalance_parity,
    test_results_independent_of_partition, test_split_off_draw, test_tombstones_and_lazy_compaction,
    test_wide_distribution_beyond_lds_window, test_scenario_update_parity, test_weight_preparation_parity,
    test_seasalt_emissions_parity, test_transport_parity, test_transport_law_1_parity, test_transport_preweight_parity,
    test_open_lateral_boundaries, test_transport_makes_room_among_tombstones,
    test_transport_slabs_match_whole_domain, test_transport_blocks_cut_in_x_and_y, test_dry_deposition_parity)
from tests.test_gpu_timestep import test_partmc_timestep_loop  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[8, 3])
def carried(request, monkeypatch):
    """8: the most a slot holds; 3: lists are cut (every merge of two longer ones)."""
    monkeypatch.setattr(parity, "N_COMP_MAX", request.param)
    return request.param


def test_lists_are_what_the_histogram_says(wp, oracle, carried, sampler_variant):
    """Size-independent properties after coagulating steps on a block too big for the oracle to
    follow cheaply: an untruncated list has exactly n_orig(s) entries of source s, its earliest
    and latest create times are the particle's least / greatest create time, and a truncated one
    is full."""
    n_cells, n_part = 96, 1500
    blk, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, 1, 1, 11, 4, n_source=3,
                                         num_conc_scale=30.0)
    for step in range(4):
        ns, nc, st = blk.mc_coag(4, 600.0, 11, step)
        blk.aero_state_rebalance(True, True, 11, step)
    dn, cm = blk.download_cells(), blk.download_components()
    K = carried
    tot = dn["n_orig"].sum(0)
    assert (tot > 1).sum() > 500 and (K > 3 or (tot > K).any())
    assert np.array_equal(cm["comp_len"], np.minimum(tot, K))
    whole = tot <= K
    for s in range(3):
        assert np.array_equal((cm["comp_source"][whole] == s).sum(1), dn["n_orig"][s][whole])
    t = np.where(cm["comp_source"] >= 0, cm["comp_time"], np.nan)
    assert np.array_equal(np.nanmin(t[whole], 1), dn["least_ct"][whole])
    assert np.array_equal(np.nanmax(t[whole], 1), dn["greatest_ct"][whole])
    cut = ~whole
    if cut.any():
        assert (np.nanmin(t[cut], 1) >= dn["least_ct"][cut]).all()
        assert (np.nanmax(t[cut], 1) <= dn["greatest_ct"][cut]).all()
    blk.close()


def test_upload_without_lists_rebuilds_them_from_the_histogram(wp, carried):
    K = carried
    blk = wp.CellBlock(2, 64, 1, n_source=3, n_comp_max=K)
    blk.set_aero_data(np.array([1800.0]))
    po = np.array([0, 3, 5], np.int64)
    n_orig = np.array([[1, 0, 2, 0, 4], [0, 1, 1, 0, 3], [0, 0, 1, 2, 5]], np.int32)
    lct = np.array([1.0, 2.0, 3.0, 4.0, 5.0]); gct = lct + 10
    blk.upload_cells(0, po, np.full((1, 5), 1e-21), None, np.arange(5, dtype=np.int64), lct, gct, n_orig)
    cm = blk.download_components()
    for i in range(5):
        want = [s for s in range(3) for _ in range(n_orig[s, i])][:K]
        assert cm["comp_len"][i] == len(want)
        assert cm["comp_source"][i, :len(want)].tolist() == want and (cm["comp_source"][i, len(want):] == -1).all()
        assert cm["comp_time"][i, :len(want)].tolist() == ([lct[i]] + [gct[i]] * (len(want) - 1))
    blk.close()


def test_components_need_a_source_table(wp):
    with pytest.raises(wp.PmcxError):
        wp.CellBlock(1, 64, 1, n_source=0, n_comp_max=4)
    with pytest.raises(wp.PmcxError):
        wp.CellBlock(1, 64, 1, n_source=2, n_comp_max=9)
    blk = wp.CellBlock(1, 64, 1, n_source=2)
    with pytest.raises(wp.PmcxError):
        blk.download_components()
    blk.close()


def test_column_file_carries_the_lists(wp, oracle, tmp_path, carried):
    """f3 with a5: output_column writes aero_component_* from the lists on the device, a fresh block
    restarted from the file holds the same lists (wrf_pmc_driver.F90:1421-1437 -> wrf_pmc_init.F90
    restart reader), and a particle of more constituents than a slot holds gets the rest from its
    histogram."""
    from wrf_partmc_amd import ncio
    K = carried
    blk, cells, info = parity.setup_pair(wp, oracle, 3, 800, 2, 2, 21, 4, cap=1024, n_source=3, num_conc_scale=20.0)
    for step in range(3):
        blk.mc_coag(4, 3000.0, 21, step)
    path = blk.output_column(str(tmp_path / "col"), 1, 2, 3, 0, 3, ["SO4", "OC"])
    dn, cm = blk.download_cells(), blk.download_components()
    col = ncio.read_column(path)
    tot = dn["n_orig"].sum(0)
    assert np.array_equal(col["comp_len"], tot) and (tot > 1).sum() > 100
    for i in np.flatnonzero(tot > 1)[:400]:
        k = int(cm["comp_len"][i])
        assert np.array_equal(col["comp_source"][i, :k], cm["comp_source"][i, :k])
        assert np.array_equal(col["comp_time"][i, :k], cm["comp_time"][i, :k])
    b2 = wp.CellBlock(3, 1024, 2, n_class=2, n_source=3, n_comp_max=K)
    b2.set_aero_data(info["dens"]); b2.set_env(280.0, 9e4)
    b2.restart_column(path, 0)
    cm2 = b2.download_components()
    for k in ("comp_len", "comp_source", "comp_time"):
        assert np.array_equal(cm2[k], cm[k]), k
    blk.close(); b2.close()
