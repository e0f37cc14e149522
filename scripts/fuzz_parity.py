#!/usr/bin/env python3
"""Randomised parity runs (not part of the test suite): mc_coag + rebalance on random small blocks --
kernel, classes, species, concentration, time step, sampler form and component-list length drawn at
random -- against the oracle, every field bit for bit (tests/parity.run_pair).
  python scripts/fuzz_parity.py [n_runs] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity  # noqa: E402
from wrf_partmc_amd import host  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
events = 0
for run in range(n_runs):
    kernel = int(rng.choice([4, 4, 4, 1, 101, 2, 3]))
    n_class = int(rng.choice([1, 1, 2, 3, 10, 40]))
    n_spec = int(rng.choice([1, 1, 2, 5, 20]))
    n_cells = int(rng.choice([1, 3, 64, 65, 130]))
    n_part = int(rng.choice([50, 300, 1000, 2500]))
    # (None: the library's choice -- with several classes the screen + list sweep; (5, 1), (5, 2): that with a
    # cell's list cut to 64 / 256 entries: many rounds; (4, 0, 0): the cells of a workgroup grouped by state)
    variant = [None, (1, 0, 0), (2, 0, 0), (3, 0, 0), (8, 0, 0), (16, 0, 0), (64, 0, 0), None, (5, 1, 0), (5, 2, 0),
               (4, 0, 0)][rng.randint(11)]
    if n_class >= 10 and variant is not None and variant[0] in (8, 16):
        variant = (64, 0, 0)      # (the list state of eight or four such cells does not fit a workgroup's LDS)
    kc = int(rng.choice([0, 0, 2, 8]))
    scale = float(10 ** rng.uniform(0, 2.2))
    del_t = float(10 ** rng.uniform(1.5, 3.5))
    coarse = bool(rng.randint(2)) and kernel in (1, 101)
    host.SAMPLER_VARIANT = variant
    parity.N_COMP_MAX = kc
    kw = dict(n_cells=n_cells, n_part=n_part, n_spec=n_spec, n_class=n_class, steps=int(rng.randint(1, 4)),
              seed=int(rng.randint(1 << 30)), kernel=kernel, del_t=del_t, rebalance=bool(rng.randint(2)),
              num_conc_scale=scale, coarse=coarse, cap=int(np.ceil(n_part * 2.2 / 64) * 64))
    res = parity.run_pair(**kw)
    if not res["bitwise_equal"] and res["status_or"] & 3 == 0:
        print("MISMATCH", kw, "variant", variant, "n_comp_max", kc, res["first_mismatch"])
        sys.exit(1)
    events += res["n_coag_total"]
    if "blk" in res:
        res["blk"].close()
    print(f"run {run}: ok  kernel {kernel} classes {n_class} species {n_spec} cells {n_cells} x {n_part} "
          f"variant {variant} lists {kc} events {res['n_coag_total']} status {res['status_or']}", flush=True)
print(f"{n_runs} runs, {events} events, {time.time() - t0:.0f} s: all equal")
