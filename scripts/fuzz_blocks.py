#!/usr/bin/env python3
"""Randomised runs of the cut-domain transport (not part of the test suite): the case of
tests/test_gpu_parity.py::test_transport_blocks_cut_in_x_and_y -- a domain as bx x by blocks with
ghost rows and columns, preweight terms and movers through the four sides, every real cell equal to
pmcx_transport on the whole domain -- with the grid, the cut, levels, classes, populations,
probabilities and list length at random.   python scripts/fuzz_blocks.py [n_runs] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from tests import parity, test_gpu_parity as tp  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0, moved, full = time.time(), 0, 0
for run in range(n_runs):
    periodic = bool(rng.randint(2))
    while True:
        bx, by, rx, ry = (int(rng.randint(1, 5)) for _ in range(4))
        nx, ny = bx * rx, by * ry
        # (a periodic direction two cells wide must not be cut: pmcx.h)
        if nx * ny <= 80 and nx >= 3 and ny >= 3:
            break
    kw = dict(nz=int(rng.choice([1, 2, 3, 7])), n_class=int(rng.randint(1, 4)), n_spec=int(rng.randint(1, 3)),
              n_part=int(rng.choice([100, 400, 700])), seed=int(rng.randint(1 << 30)),
              prob=float(rng.choice([0.02, 0.08, 0.2])), ideal=float(rng.choice([0.45, 1.0, 3.0])),
              n_steps=int(rng.randint(1, 4)), cap_factor=8.0, law=int(rng.randint(2)))
    parity.N_COMP_MAX = int(rng.choice([0, 0, 3, 8]))
    print(f"run {run}: periodic {periodic} shape {(nx, ny, bx, by)} lists {parity.N_COMP_MAX} {kw}", flush=True)
    try:
        m, _ = tp.blocks_cut_in_x_and_y(wp, periodic, (nx, ny, bx, by), **kw)
    except AssertionError as e:
        if "is full" not in str(e):
            raise
        print("   ", e); full += 1; continue
    moved += m
print(f"{n_runs} runs ({full} with a full cell), {moved} movers, {time.time() - t0:.0f} s: all equal")
