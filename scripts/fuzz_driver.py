#!/usr/bin/env python3
"""Randomised runs of the two Fortran drop-in drivers (not part of the test suite): the case of
tests/test_gpu_driver.py -- WRF steps through partmc_timestep_pmcx and wrf_pmc_trans_driver_pmcx,
the oracle through the same loop, every particle, component list, weight and gas compared bit for
bit -- with grid shape, particle count, number of steps, boundaries, sea-salt option and seed at
random.   python scripts/fuzz_driver.py [n_runs] [seed] [max_steps] [headroom] [auto_grow]
headroom: slots per cell / ideal count the Fortran side STARTS with (default: the shim's 3; the oracle
keeps three times the ideal count, the results do not depend on it); auto_grow 0: the block does not
grow (a full cell stops the run, as until round 4)."""
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from oracle import pmco_py as oracle  # noqa: E402
from tests import test_gpu_driver as drv, wrf_case  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if len(sys.argv) > 4:
    os.environ["PMCX_TEST_HEADROOM"] = sys.argv[4]
if len(sys.argv) > 5:
    os.environ["PMCX_TEST_AUTO_GROW"] = sys.argv[5]
assert os.path.exists(drv.EXE), "build wrf-partmc_amd/fortran first (make)"
t0 = time.time()
full = 0
for run in range(n_runs):
    periodic = bool(rng.randint(2))
    lo = 2 if periodic else 3
    kw = dict(nx=int(rng.randint(lo, 7)), nz=int(rng.randint(2, 7)), ny=int(rng.randint(lo, 7)),
              n_part=int(rng.choice([60, 150, 300, 500])), n_steps=int(rng.randint(1, max_steps)), periodic=periodic,
              seed=int(rng.randint(1 << 30)), sea_salt_opt=int(rng.randint(0, 3)))
    print(f"run {run}: {kw}", flush=True)
    case = wrf_case.Case(**kw)
    with tempfile.TemporaryDirectory() as d:
        try:
            steps, final = drv.run_case(case, pathlib.Path(d), wp, oracle)
        except AssertionError as e:
            if "ran out of particle slots" not in str(e):
                raise
            msg = str(e)
            at = msg.index("ran out of particle slots")
            print("   the driver stopped: a cell " + msg[at:at + 70].split(":")[0] + " (the shim's capacity rule)", flush=True)
            full += 1
            continue
    print(f"   moved {sum(s['n_moved'] for s in steps)} coagulated {sum(s['n_coag'] for s in steps)} "
          f"deposited {sum(s['n_deposited'] for s in steps)} particles {sum(f['n'] for f in final)}", flush=True)
print(f"{n_runs} runs ({full} stopped by a full cell), {time.time() - t0:.0f} s: all equal"
      + (f" (headroom {sys.argv[4]})" if len(sys.argv) > 4 else ""))
