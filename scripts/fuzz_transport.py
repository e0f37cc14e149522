#!/usr/bin/env python3
"""Randomised parity runs of the transport chain (not part of the test suite): preweight -> transport
(periodic or open lateral boundaries) -> dry deposition -> rebalance -> mc_coag on random grids -- shape, classes,
species, probabilities (also saturated rows, zero rows, empty cells), weight spread, list length at
random -- against the oracle, every field bit for bit.   python scripts/fuzz_transport.py [n_runs] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from oracle import pmco_py as oracle  # noqa: E402
from tests import parity  # noqa: E402
from wrf_partmc_amd import host  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t0 = time.time()
moved = 0
for run in range(n_runs):
    periodic = bool(rng.randint(2))
    nx, ny = (int(rng.randint(1, 6)), int(rng.randint(1, 6))) if periodic else (int(rng.randint(3, 7)), int(rng.randint(3, 7)))
    nz = int(rng.choice([1, 2, 5, 30, 61]))
    n_class, n_spec = int(rng.choice([1, 2, 3, 3, 12])), int(rng.choice([1, 2]))
    n_part = int(rng.choice([40, 200, 600]))
    n = nx * nz * ny
    if n * n_part > 400000:
        nz = 2; n = nx * nz * ny
    kc = int(rng.choice([0, 0, 3, 8]))
    host.SAMPLER_VARIANT = [None, (1, 0, 0), (3, 0, 0)][rng.randint(3)]
    law = int(rng.randint(2))            # pmcx_set_transport_law
    parity.N_COMP_MAX = kc
    seed = int(rng.randint(1 << 30))
    blk, cells, info = parity.setup_pair(wp, oracle, n, n_part, n_spec, n_class, seed, 4, cap=int(np.ceil(n_part * 3.0 / 64) * 64),
                                         n_source=2, num_conc_scale=float(10 ** rng.uniform(0, 1.5)))
    # weights spread by up to 10x between cells, some cells empty
    spread = float(rng.choice([0.5, 0.5, 1.2]))       # (1.2: ratios down to 4e-3 -- movers and emitters overlap)
    mag = info["mag"] * 10 ** rng.uniform(-spread, spread, (n, 1))
    blk.set_weights(mag)
    for c, oc in enumerate(cells):
        oc.mag[:] = mag[c]
    scale_h, scale_v = 10 ** rng.uniform(-3, -0.3), 10 ** rng.uniform(-3, -0.5) / max(nz, 1)
    ph = rng.uniform(0, scale_h, (n, n_class, 4)); pv = rng.uniform(0, scale_v, (n, n_class, nz))
    ph[rng.rand(n) < 0.1] = 0.0                                   # rows that send nothing sideways
    sat = rng.rand(n) < 0.05
    ph[sat] = rng.uniform(0.2, 0.5, (int(sat.sum()), n_class, 4))  # saturated tables (cumulative sum clipped at 1)
    cv = rng.uniform(0.5e9, 2e9, n)
    n_ideal = np.full((n, n_class), n_part / n_class * float(rng.choice([0.3, 1.0, 3.0])))
    blk.set_n_part_ideal(n_ideal)
    if not periodic:
        blk.set_transport_lateral(True)
    blk.set_transport_law(law); oracle.set_transport_law(law)
    print(f"run {run}: {nx} x {nz} x {ny} periodic {periodic} classes {n_class} species {n_spec} particles {n_part} "
          f"lists {kc} variant {host.SAMPLER_VARIANT} law {law} seed {seed}", flush=True)
    next_id = np.array([oc.next_id for oc in cells], dtype=np.int64)
    surface = ((np.arange(n) // nx) % nz == 0).astype(np.int32)
    for step in range(int(rng.randint(1, 4))):
        ex, nr = blk.transport_preweight(nx, nz, ny, ph, pv, cv, seed, step, periodic, periodic)
        oex, onr = oracle.transport_preweight(cells, nx, nz, ny, ph, pv, cv, n_ideal, seed, 0, step, periodic, periodic)
        assert np.array_equal(ex, oex) and np.array_equal(nr, onr), ("preweight", run, step)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, "after preweight", msg)
        nm, na = blk.transport(nx, nz, ny, ph, pv, cv, seed, step, periodic, periodic)
        st, onm, ona = oracle.transport(cells, nx, nz, ny, ph, pv, cv, seed, 0, step, next_id, periodic, periodic,
                                        lateral=not periodic)
        assert np.array_equal(nm, onm) and np.array_equal(na, ona), ("transport", run, step)
        st_dev = int(np.bitwise_or.reduce(blk.coag_counters()[2]))
        assert st_dev == st, ("status", run, step, st_dev, st)
        if st:
            print(f"   status {st:#x} at step {step}")
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, msg)
        assert np.array_equal(blk.get_next_id(), next_id)
        # dry deposition on the surface cells, then the rebalance (n_ideal 0.3 x and 3 x the population)
        ra, us = rng.uniform(20, 200, n), rng.uniform(0.1, 0.6, n)
        height, rrho = rng.uniform(30, 80, n), 1 / rng.uniform(1.0, 1.25, n)
        alpha, big_a = rng.choice([1.2, 100.0], n), rng.choice([0.002, 9.999], n)
        nr, bad = blk.dry_deposition(1500.0, ra, us, alpha, big_a, height, rrho, seed=seed, step=step, surface=surface)
        for c, oc in enumerate(cells):
            if surface[c]:
                assert oc.dry_deposition(1500.0, seed, c, step, ra[c], us[c], alpha[c], big_a[c], height[c],
                                         rrho[c]) == (int(nr[c]), int(bad[c])), ("deposition", run, step, c)
        blk.aero_state_rebalance(True, True, seed, step)
        for c, oc in enumerate(cells):
            _, next_id[c] = oc.rebalance(n_ideal[c], True, True, seed, oc.cell_id, step, int(next_id[c]))
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, "after deposition and rebalance", msg)
        assert np.array_equal(blk.get_next_id(), next_id)
        ns, nc, s2 = blk.mc_coag(4, 300.0, seed, step)
        for c, oc in enumerate(cells):
            assert oc.mc_coag(4, 300.0, seed, oc.cell_id, step) == (int(s2[c]), int(ns[c]), int(nc[c])), (run, step, c)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, "after mc_coag", msg)
        moved += int(nm.sum())
    blk.close()
    oracle.set_transport_law()
print(f"{n_runs} runs, {moved} movers, {time.time() - t0:.0f} s: all equal")
