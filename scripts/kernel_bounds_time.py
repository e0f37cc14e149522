#!/usr/bin/env python3
"""Cost of the kernel bounds (k_kernel_bounds) on 122 880 cells: whole tables
(pmcx_update_kernel_bounds), the populated range + margin that pmcx_mc_coag computes by itself
after a change of environment, and the look that finds every cell's bounds still valid."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wrf_partmc_amd as wp  # noqa: E402
from wrf_partmc_amd.data import idealised_modes  # noqa: E402

n, N = 122880, 1024
for S, dens in ((1, [1770.0]), (2, [1000.0, 2600.0])):
    blk = wp.CellBlock(n, N, S)
    blk.set_aero_data(dens)
    T, P = np.linspace(210, 300, n), np.linspace(2e4, 1e5, n)
    blk.set_env(T, P, np.full(n, 0.5))
    blk.set_weights(np.full((n, 1), 1e10 / N))
    blk.set_next_id(np.arange(n, dtype=np.int64) << 32)
    blk.generate_lognormal(N, idealised_modes(S, 1), seed=0)
    blk.update_kernel_bounds(4); blk.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        blk.update_kernel_bounds(4)
    blk.synchronize()
    whole = (time.perf_counter() - t) / 3 * 1e3
    blk.profile_enable(True)
    out = {}
    for label, change in (("after set_env", True), ("still valid", False)):
        blk.profile_reset()
        for k in range(3):
            if change:
                blk.set_env(T + 0.5 * (k + 1), P, np.full(n, 0.5))
            else:
                blk.invalidate_sort()          # marks the populations as possibly changed
            blk.mc_coag(4, 60.0, 0, k, counters=False)
        blk.synchronize()
        ms, cnt = blk.profile_get("kernel_bounds")
        out[label] = ms / max(cnt, 1)
    print(f"n_spec {S}: whole tables {whole:.2f} ms | populated range + 8 bins {out['after set_env']:.2f} ms"
          f" | look only {out['still valid']:.3f} ms   ({n} cells)")
    blk.close()
