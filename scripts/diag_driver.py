#!/usr/bin/env python3
"""Diagnosis aid: one case of scripts/fuzz_driver.py at growing step counts; where the driver and the
oracle part, which cells and slots, and whether the cell holds the same particles in another order."""
import os, pathlib, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from oracle import pmco_py as oracle  # noqa: E402
from tests import test_gpu_driver as drv, wrf_case  # noqa: E402

kw = eval(sys.argv[1])
for n_steps in range(1, kw["n_steps"] + 1):
    case = wrf_case.Case(**dict(kw, n_steps=n_steps))
    with wp.CellBlock(1, 64, case.n_spec, n_class=case.n_class) as blk:
        v_edge, v_samp = blk.bin_tables()
        const_k, beta = blk.cfg.const_k, blk.cfg.additive_beta
        max_events, max_appends = blk.cfg.max_events, blk.cfg.max_appends
    cap = max(64, int(3.0 * case.num_particles), case.n_part, int(np.ceil(3.0 * case.n_ideal.sum(1).max())))
    cap = (cap + 63) // 64 * 64
    cells = case.make_cells(oracle, v_edge, cap, max_events, max_appends)
    with tempfile.TemporaryDirectory() as d:
        d = pathlib.Path(d)
        (d / "s").write_bytes(case.state_bytes(cells)); (d / "f").write_bytes(case.fields_bytes())
        import subprocess
        run = subprocess.run([drv.EXE, str(d / "s"), str(d / "f"), str(d / "o")], capture_output=True, text=True)
        if run.returncode != 0:
            print(f"n_steps {n_steps}: the driver stopped:", (run.stdout[-300:] + run.stderr[-600:]).strip().splitlines()[-3:])
            break
        steps, final = case.parse_output((d / "o").read_bytes())
    for oc in cells:
        oc.refresh_caches(); oc.sort_valid[0] = 0
    counts, gas = case.run_oracle(oracle, cells, v_samp, const_k, beta, steps, wp.host)
    bad = []
    for c, (oc, fc) in enumerate(zip(cells, final)):
        n = int(oc.n_part[0])
        if fc["n"] != n:
            bad.append((c, "count", fc["n"], n)); continue
        if not np.array_equal(fc["vol"], oc.vol[:, :n]) or not np.array_equal(fc["id"], oc.id[:n]):
            d_ = np.flatnonzero((fc["vol"] != oc.vol[:, :n]).any(0) | (fc["id"] != oc.id[:n]))
            perm = np.array_equal(np.sort(fc["id"]), np.sort(oc.id[:n]))
            bad.append((c, "slots", d_[:6].tolist(), len(d_), "of", n, "same ids" if perm else "other ids",
                        "dev", fc["id"][d_[:3]].tolist(), fc["vol"][:, d_[0]].tolist(), "orc", oc.id[d_[:3]].tolist(),
                        oc.vol[:, d_[0]].tolist()))
        if not np.array_equal(fc["mag"], oc.mag):
            bad.append((c, "mag", fc["mag"].tolist(), oc.mag.tolist()))
    cnt = [int((steps[s]["n_parts"] != counts[s]).sum()) for s in range(n_steps)]
    mags = np.array([oc.mag for oc in cells]) / case.mag
    print(f"   cap {cap}: largest cell per step {[int(x.max()) for x in counts]}; weights relative to the start: "
          f"min {mags.min():.3f} max {mags.max():.3f}")
    print(f"n_steps {n_steps}: cells that differ {len(bad)}; count mismatches per step {cnt}", flush=True)
    for b in bad[:6]:
        print("    ", b)
    if bad:
        break
