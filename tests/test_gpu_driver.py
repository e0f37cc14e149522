"""-m gpu: the two drop-in drivers of the Fortran shim -- partmc_timestep_pmcx with the argument
list of partmc_timestep (interface/wrf_pmc_driver.F90:46-87) and wrf_pmc_trans_driver_pmcx with that
of wrf_pmc_trans_driver (interface/wrf_pmc_trans.F90:37-76) -- driven by flang the way
WRFV3/frame/module_integrate.F:409-421 drives the reference, seven WRF steps of 20 s with
partmc_chem_dt = 60 s: coagulation on chemistry steps only, gas and aerosol emissions (the series
move to their next time level mid-run), dilution, sea salt over the ocean columns, rebalance;
transport probabilities from WRF-shaped fields, preweight, sampling, open lateral boundaries with
their refill (or a doubly periodic domain), dry deposition.  Nothing but those two calls touches the
device between the first step and the final partmc_sync_host_pmcx.  The CPU oracle runs the same
loop cell by cell; every particle, weight and gas mixing ratio must agree bit for bit."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import wrf_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "wrf-partmc_amd", "fortran")
EXE = os.path.join(FDIR, "test_driver")


def _build():
    if not os.path.exists(EXE) and shutil.which("make") and os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        subprocess.call(["make", "-C", FDIR, "-s"])
    if not os.path.exists(EXE):
        pytest.skip("flang or the prebuilt test_driver is not available")


def run_case(case, tmp_path, wp, oracle):
    """The Fortran driver and the oracle through the same case; every particle, list, weight and
    gas compared.  Returns (steps, final) of the driver's dump.  (Also scripts/fuzz_driver.py.)"""
    with wp.CellBlock(1, 64, case.n_spec, n_class=case.n_class) as blk:      # the product's tables and defaults
        v_edge, v_samp = blk.bin_tables()
        const_k, beta = blk.cfg.const_k, blk.cfg.additive_beta
        max_events, max_appends = blk.cfg.max_events, blk.cfg.max_appends
    # the shim's capacity rule (pmcx_upload_states): the headroom (3) times the ideal count, its caller's 3 x
    # num_particles, rounded up to 64
    cap = max(64, int(3.0 * case.num_particles), case.n_part, int(np.ceil(3.0 * case.n_ideal.sum(1).max())))
    cap = (cap + 63) // 64 * 64
    cells = case.make_cells(oracle, v_edge, cap, max_events, max_appends)
    fstate, ffields, fout = tmp_path / "state.bin", tmp_path / "fields.bin", tmp_path / "out.bin"
    fstate.write_bytes(case.state_bytes(cells))
    ffields.write_bytes(case.fields_bytes())
    run = subprocess.run([EXE, str(fstate), str(ffields), str(fout)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    steps, final = case.parse_output(fout.read_bytes())

    for oc in cells:                   # an upload recomputes the caches and invalidates the sort
        oc.refresh_caches()
        oc.sort_valid[0] = 0
    counts, gas = case.run_oracle(oracle, cells, v_samp, const_k, beta, steps, wp.host)
    for s in range(case.n_steps):
        assert np.array_equal(steps[s]["n_parts"], counts[s]), (
            s, np.flatnonzero(steps[s]["n_parts"] != counts[s])[:8], steps[s]["n_parts"][:8], counts[s][:8])
    for c, (oc, fc) in enumerate(zip(cells, final)):
        n = int(oc.n_part[0])
        assert fc["n"] == n, c
        assert np.array_equal(fc["vol"], oc.vol[:, :n]), c
        assert np.array_equal(fc["id"], oc.id[:n]), c
        assert np.array_equal(fc["wclass"], oc.wclass[:n]), c
        assert np.array_equal(fc["n_orig"], oc.n_orig[:, :n]), c
        assert np.array_equal(fc["ct"][:, 0], oc.lct[:n]) and np.array_equal(fc["ct"][:, 1], oc.gct[:n]), c
        assert np.array_equal(fc["mag"], oc.mag), c
        # component(:) (driver:1421-1437): source and create time of every constituent, through
        # coagulation, doubling, transport copies, boundary copies and refills, emissions, sea salt
        tot = oc.n_orig[:, :n].sum(0)
        for i in range(n):
            k, lst = int(oc.comp_n[i]), fc["comp"][i]
            assert len(lst) == tot[i] and k == min(int(tot[i]), 8), (c, i)
            assert np.array_equal(lst["s"][:k] - 1, oc.comp_src[i, :k]), (c, i)
            assert np.array_equal(lst["t"][:k], oc.comp_time[i, :k]), (c, i)
            assert (lst["t"][k:] == oc.gct[i]).all(), (c, i)
        assert np.array_equal(fc["gas"].view(np.uint64), gas[c].view(np.uint64)), c
    return steps, final


@pytest.mark.parametrize("periodic", [False, True])
def test_wrf_steps_through_the_two_drivers(tmp_path, wp, oracle, periodic):
    _build()
    case = wrf_case.Case(periodic=periodic, seed=91 if not periodic else 57)
    steps, final = run_case(case, tmp_path, wp, oracle)
    assert any(len(l) >= 2 for fc in final for l in fc["comp"])           # coagulated particles: lists of two constituents
    # the case exercises what it claims to
    tot = lambda k: sum(st[k] for st in steps)
    chem = [st["n_coag"] for st in steps]
    assert chem[0] > 0 and chem[3] > 0 and chem[6] > 0           # mc_coag on steps 0, 3, 6 ...
    assert chem[1] == chem[2] == chem[4] == chem[5] == 0         # ... and on no other (driver:189)
    assert tot("n_moved") > 5 * case.n_cells and tot("n_deposited") > 0
    ids = np.concatenate([fc["id"] for fc in final])
    assert len(np.unique(ids)) == len(ids)                       # particle ids stay unique
    created = ids[ids < (1 << 40)]                               # (the initial ids start at 2**40)
    assert len(created) > 50                                     # emission, sea salt, copies, refills
    if not periodic:
        # cells outside partmc_timestep's loop nest: no gas update ...
        out = ~case.inside
        assert np.array_equal(np.array([fc["gas"] for fc in final])[out], case.gas0[out])
        assert not np.array_equal(np.array([fc["gas"] for fc in final])[case.inside], case.gas0[case.inside])
