#!/usr/bin/env python3
"""Achieved HBM GB/s of every full-population kernel at a size past the 256 MiB
Infinity Cache (default 60000 cells x 10^4 particles; --n-spec species).
Algorithmic bytes per particle as in DESIGN.md section 5.  Writes one JSON line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=60000)
ap.add_argument("--particles", type=int, default=10000)
ap.add_argument("--n-spec", type=int, default=1)
ap.add_argument("--n-class", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
S, C, n, N = a.n_spec, a.n_class, a.cells, a.particles
cap = (N + 63) // 64 * 64
from wrf_partmc_amd.data import aero_data_arrays
dens, kappa, _, names = aero_data_arrays()
i_water = S - 1 if S > 1 else -1
blk = wp.CellBlock(n, cap, S, n_class=C, n_source=0, i_water=i_water, max_appends=128)
blk.set_aero_data(dens[:S] if S < 20 else dens, kappa=(kappa[:S] if S < 20 else kappa))
blk.set_env(np.linspace(290, 220, n), np.linspace(101325, 25000, n), np.full(n, 0.8))
blk.set_weights(np.full((n, C), 1e10 / N))
blk.set_n_part_ideal(np.full((n, C), N / C))
vf = np.zeros(S); vf[0] = 1.0
modes = [dict(frac=1.0, gm_radius=2e-8, gsd_log10=0.25, vol_frac=vf, wclass=k % C, source=0) for k in range(C)]
blk.generate_lognormal(N, modes, seed=0)
blk.synchronize()
P = n * N
out = {"cells": n, "particles_per_cell": N, "n_spec": S, "n_class": C, "hbm_peak_gbs": 8000.0, "kernels": {}}


def timed(name, fn, bytes_per_particle, reps=a.reps):
    fn(); blk.synchronize()
    blk.profile_enable(True); blk.profile_reset()
    for _ in range(reps):
        fn()
    blk.synchronize()
    ms, k = blk.profile_get(name)
    blk.profile_enable(False)
    ms /= max(k, 1)
    gbs = P * bytes_per_particle / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out["kernels"][name] = {"ms": ms, "algorithmic_bytes_per_particle": bytes_per_particle,
                            "achieved_gbs": gbs, "frac_of_peak": gbs / 8000.0}


def resort():            # a sort of every cell: lists that still hold are skipped
    blk.invalidate_sort()
    blk.sort_only()


timed("bin_sort", resort, 12 + (4 if C > 1 else 0))
timed("diagnostics", blk.diagnostics, 8 * S + 4)
if S > 1:
    timed("condense", lambda: blk.condense(300.0, 4), 16 * S + 4 + 24)
if C > 1:
    timed("rebalance", lambda: blk.aero_state_rebalance(True, True, 0, 0), 4)
print(json.dumps(out))
