#!/usr/bin/env python3
"""Randomised parity runs of the adds with weight preparation (not part of the test suite):
scenario update (emission, dilution, background), boundary refill, rebalance, mc_coag on small
cells with ideal counts far off their populations, small capacities, every stage against the oracle.
   python scripts/fuzz_prep.py [n_runs] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from oracle import pmco_py as oracle  # noqa: E402
from tests import parity  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t0 = time.time()
for run in range(n_runs):
    n_cells, n_class, n_spec = 24, int(rng.randint(1, 4)), int(rng.randint(1, 3))
    n_part = int(rng.choice([40, 60, 150, 400]))
    cap = int(np.ceil(n_part * float(rng.choice([8.0, 14.0, 24.0])) / 64) * 64)
    parity.N_COMP_MAX = int(rng.choice([0, 0, 3, 8]))
    seed = int(rng.randint(1 << 30))
    flags = (bool(rng.randint(4)), bool(rng.randint(4)))
    print(f"run {run}: classes {n_class} species {n_spec} particles {n_part} cap {cap} lists {parity.N_COMP_MAX} "
          f"flags {flags} seed {seed}", flush=True)
    blk, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, seed, 4, cap=cap, n_source=3)
    mag = info["mag"]
    n_ideal = n_part / n_class * 10 ** rng.uniform(-0.6, 0.6, (n_cells, n_class))
    blk.set_n_part_ideal(n_ideal)
    blk.set_weight_preparation(True, *flags)
    for c, oc in enumerate(cells):
        oc.set_weight_preparation(n_ideal[c], *flags)
    vf = lambda: (lambda v: (v / v.sum()).tolist())(rng.uniform(0.1, 1, n_spec))
    em = [dict(gm_radius=2e-8, gsd_log10=0.2, vol_frac=vf(), wclass=int(rng.randint(n_class)), source=1),
          dict(gm_radius=1e-7, gsd_log10=0.1, vol_frac=vf(), wclass=int(rng.randint(n_class)), source=2)]
    bg = [dict(gm_radius=5e-8, gsd_log10=0.3, vol_frac=vf(), wclass=int(rng.randint(n_class)), source=0)]
    full = False
    for step in range(int(rng.randint(2, 8))):
        em_rate = mag[:, [m["wclass"] for m in em]] * 10 ** rng.uniform(-1.5, 0.3, (n_cells, 2)) * n_part / 60.0
        bg_conc = mag[:, [m["wclass"] for m in bg]] * n_part * 10 ** rng.uniform(-1, 0.5, (n_cells, 1))
        dil = rng.uniform(0, 5e-3, n_cells) * (rng.rand(n_cells) < 0.7)
        hgt = rng.uniform(30, 300, n_cells); hgt_old = hgt * rng.choice([0.9, 1.0, 1.1], n_cells)   # the layer grew, stayed, shrank
        ne, ni, no = blk.scenario_update_aero_state(60.0, 60.0 * (step + 1), seed, step, em, em_rate, dil, bg, bg_conc, hgt, hgt_old)
        st_o = 0
        for c, oc in enumerate(cells):
            oe = [dict(m, num_conc=em_rate[c, k]) for k, m in enumerate(em)]
            ob = [dict(m, num_conc=bg_conc[c, k]) for k, m in enumerate(bg)]
            st, oc.next_id, one, oni, ono = oc.scenario_update(60.0, 60.0 * (step + 1), seed, oc.cell_id, step, oe, dil[c], ob,
                                                               oc.next_id, hgt[c], hgt_old[c])
            st_o |= st
            assert (one, oni, ono) == (int(ne[c]), int(ni[c]), int(no[c])), ("scenario", run, step, c, (one, oni, ono),
                                                                              (int(ne[c]), int(ni[c]), int(no[c])))
        st_d = int(np.bitwise_or.reduce(blk.coag_counters()[2]))
        assert st_d == st_o, ("status", run, step, st_d, st_o)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, "after the scenario update", msg)
        assert np.array_equal(blk.get_next_id(), [oc.next_id for oc in cells])
        assert np.array_equal(blk.get_weights(), np.array([oc.mag for oc in cells]))
        if st_o:
            print(f"   status {st_o:#x} at step {step}"); full = True; break
        lst = np.sort(rng.choice(n_cells, 6, replace=False)).astype(np.int32); thr = rng.randint(0, 2, 6).astype(np.int32)
        conc = np.array([[n_ideal[c, bg[0]["wclass"]] * blk.get_weights()[c, bg[0]["wclass"]] * 10 ** rng.uniform(-1.0, 0.4)]
                         for c in lst])
        na, nr = blk.boundary_resample(lst, thr, bg, conc, 1.0, 0.0, seed, step)
        for q, c in enumerate(lst):
            st, cells[c].next_id, ona, onr = cells[c].boundary_resample(int(thr[q]), [dict(bg[0], num_conc=conc[q, 0])], 1.0,
                                                                        0.0, seed, cells[c].cell_id, step, cells[c].next_id)
            st_o |= st
            assert (ona, onr) == (int(na[q]), int(nr[q])), ("refill", run, step, c)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, "after the refill", msg)
        if st_o or int(np.bitwise_or.reduce(blk.coag_counters()[2])):
            print(f"   status at the refill, step {step}"); full = True; break
        blk.aero_state_rebalance(True, True, seed, step)
        for c, oc in enumerate(cells):
            _, oc.next_id = oc.rebalance(n_ideal[c], True, True, seed, oc.cell_id, step, oc.next_id)
        ns, nc, s2 = blk.mc_coag(4, 300.0, seed, step)
        for c, oc in enumerate(cells):
            assert oc.mc_coag(4, 300.0, seed, oc.cell_id, step) == (int(s2[c]), int(ns[c]), int(nc[c])), (run, step, c)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, (run, step, "after mc_coag", msg)
    blk.close()
print(f"{n_runs} runs, {time.time() - t0:.0f} s: all equal")
