#!/usr/bin/env python3
"""Wall time of pmcx_scenario_update_aero_state (emission + dilution + background) on a block of cells,
with the adds' weight preparation off and on, one and two weight classes.
   python scripts/time_scenario.py [n_cells] [n_part]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from wrf_partmc_amd.data import idealised_modes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
for n_class in (1, 2):
    for prep in (False, True):
        blk = wp.CellBlock(n, (int(N * 2.2) + 63) // 64 * 64, 1, n_class=n_class, n_source=2)
        blk.set_aero_data([1770.0]); blk.set_env(288.0, 1e5, 0.5)
        blk.set_weights(np.full((n, n_class), 1e10 / N)); blk.set_n_part_ideal(np.full((n, n_class), N / n_class))
        blk.set_next_id((np.arange(n, dtype=np.int64) << 32) + (1 << 24))
        modes = idealised_modes(1, 1)
        if n_class == 2:
            modes = [dict(m, frac=0.5, wclass=q) for q in range(2) for m in modes]
        blk.generate_lognormal(N, modes, seed=0)
        if prep:
            blk.set_weight_preparation(True, True, True)
        em = [dict(gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0], wclass=q % n_class, source=1) for q in range(2)]
        bg = [dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[1.0], wclass=0, source=0)]
        em_rate = np.where((np.arange(n) % 30 == 0)[:, None], 2e5, -1.0) * np.ones((n, 2))   # the ground level emits (-1: no such mode)
        bg_conc = np.full((n, 1), 1e10); dil = np.full(n, 1e-4)
        ts = []
        for step in range(6):
            blk.synchronize(); t0 = time.perf_counter()
            blk.scenario_update_aero_state(20.0, 20.0 * (step + 1), 0, step, em, em_rate, dil, bg, bg_conc)
            blk.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"classes {n_class} preparation {'on ' if prep else 'off'}: {np.median(ts[1:]):.2f} ms per call "
              f"({n} cells x {N}), particles {int(blk.aero_state_total_particles().sum())}", flush=True)
        blk.close()
