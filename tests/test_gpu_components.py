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


def test_column_file_against_the_oracle(wp, oracle, tmp_path, carried):
    """f3 held to the ORACLE, not to the device's own download: three cells through coagulating steps and a
    rebalance on both sides, the device writes its column (output_column: the layout of
    output_column_to_file_new, wrf_pmc_driver.F90:1292-1658), and the file -- walked by
    tests/cdf5_walker.py, a reader written from the format grammar that shares no code with the writer --
    must hold what the oracle's cells hold, variable by variable: part_start_index / n_parts, masses =
    vol x density, classes (1-based), ids, create times, number concentrations, the weights per level,
    next_id, and aero_component_* (lengths = the untruncated sums of the source histogram, the first
    n_comp_max entries of every list the oracle's, in its order)."""
    from tests.cdf5_walker import walk
    K = carried
    n_cells, n_part, n_spec, n_class = 3, 900, 2, 2
    blk, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, 27, 4, cap=2048, n_source=3,
                                         num_conc_scale=20.0)
    ideal = np.full(n_class, n_part / n_class)
    for step in range(3):
        ns, nc, st = blk.mc_coag(4, 3000.0, 27, step)
        for c, oc in enumerate(cells):
            assert oc.mc_coag(4, 3000.0, 27, oc.cell_id, step) == (int(st[c]), int(ns[c]), int(nc[c]))
        blk.aero_state_rebalance(True, True, 27, step)
        for oc in cells:
            _, oc.next_id = oc.rebalance(ideal, True, True, 27, oc.cell_id, step, oc.next_id)
    path = blk.output_column(str(tmp_path / "col"), 4, 5, 6, 0, n_cells, ["SO4", "BC"], time=180.0, del_t=60.0)
    dims, gatts, v = walk(path)
    n = [int(oc.n_part[0]) for oc in cells]
    po = np.concatenate([[0], np.cumsum(n)])
    assert dims["number_levels"] == n_cells and dims["aero_particle"] == po[-1] and dims["aero_species"] == n_spec
    assert dims["aero_weight_class"] == n_class and dims["aero_source"] == 3
    assert v["time"][2][()] == 180.0 and v["timestep"][2][()] == 60.0 and v["timestep_index"][2][()] == 6
    assert v["n_parts"][2].tolist() == n and v["part_start_index"][2].tolist() == (po[:-1] + 1).tolist()
    dens = info["dens"]
    cat = lambda f: np.concatenate([f(oc)[:k] for oc, k in zip(cells, n)])
    mass = np.concatenate([oc.vol[:, :k] for oc, k in zip(cells, n)], axis=1) * dens[:, None]
    assert v["aero_particle_mass"][0] == ["aero_species", "aero_particle"]
    np.testing.assert_array_equal(v["aero_particle_mass"][2], mass)
    np.testing.assert_array_equal(v["aero_particle_weight_class"][2], cat(lambda oc: oc.wclass) + 1)
    assert (v["aero_particle_weight_group"][2] == 1).all()
    np.testing.assert_array_equal(v["aero_id"][2], cat(lambda oc: oc.id))
    np.testing.assert_array_equal(v["aero_least_create_time"][2], cat(lambda oc: oc.lct))
    np.testing.assert_array_equal(v["aero_greatest_create_time"][2], cat(lambda oc: oc.gct))
    np.testing.assert_array_equal(v["aero_num_conc"][2],
                                  np.concatenate([oc.mag[oc.wclass[:k]] for oc, k in zip(cells, n)]))
    assert v["weight_magnitude"][0] == ["aero_weight_class", "aero_weight_group", "number_levels"]
    np.testing.assert_array_equal(v["weight_magnitude"][2][:, 0, :], np.array([oc.mag for oc in cells]).T)
    assert (v["weight_type"][2] == 1).all() and (v["weight_exponent"][2] == 0).all()
    assert v["next_id"][2][()] == max(oc.next_id for oc in cells)
    # component(:): lengths, start indices, owners; the entries the device carries against the oracle's lists
    tot = np.concatenate([oc.n_orig[:, :k].sum(0) for oc, k in zip(cells, n)])
    ln, start = v["aero_component_len"][2], v["aero_component_start_ind"][2]
    np.testing.assert_array_equal(ln, tot)
    np.testing.assert_array_equal(start, np.concatenate([[1], 1 + np.cumsum(tot)[:-1]]))
    assert dims["aero_components"] == tot.sum() and (tot > 1).sum() > 100
    np.testing.assert_array_equal(v["aero_component_particle_num"][2], np.repeat(np.arange(1, po[-1] + 1), tot))
    src, tim = v["aero_component_source_num"][2], v["aero_component_create_time"][2]
    o_n = cat(lambda oc: oc.comp_n); o_src = np.concatenate([oc.comp_src[:k] for oc, k in zip(cells, n)])
    o_tim = np.concatenate([oc.comp_time[:k] for oc, k in zip(cells, n)])
    checked = 0
    for i in range(int(po[-1])):
        k = int(min(o_n[i], K))
        assert o_n[i] == min(tot[i], K)
        a = int(start[i]) - 1
        assert np.array_equal(src[a:a + k], o_src[i, :k] + 1), i
        assert np.array_equal(tim[a:a + k], o_tim[i, :k]), i
        checked += k > 1
    assert checked > 100
    blk.close()
