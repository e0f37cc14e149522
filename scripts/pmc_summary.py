"""Summarise rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch."""
import csv, glob, json, os, sys
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, f"pmc_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    d["dispatches"] = max(len(v) for v in acc[k].values())
json.dump(out, open(os.path.join(root, f"pmc_{tag}_summary.json"), "w"), indent=1)
for k in sorted(out):
    if "pmcx" in k:
        print(k, {c: (f"{v:.4g}") for c, v in out[k].items()})

# HBM traffic per launch for bench.py's roofline.traffic (profiles/pmc_traffic.json).
# FETCH_SIZE/WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a
# coalesced streaming read (MI355X_MICROARCH.md, HBM): k_bin_sort's read side is such a
# stream and is calibrated here (algorithmic 8 B/particle read once == 2 x FETCH_SIZE);
# the sampler's gathers (4 to 16 bytes per lane at independent addresses) are calibrated by
# scripts/gather_probe.hip: FETCH_SIZE counts 64 B per such load, used raw.
traffic = {}
names = {"pmcx::k_bin_sort": ("bin_sort", 2.0), "pmcx::k_rebalance": ("rebalance", 1.0),
         "pmcx::k_compact": ("compact", 1.0)}
# the sampler under whichever template arguments it ran with (the one with the most traffic)
samplers = [k for k in out if k.startswith("pmcx::k_coag_lane<") or k.startswith("pmcx::k_coag_sample<")]
samplers = [k for k in samplers if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]]
if samplers:
    names[max(samplers, key=lambda k: out[k]["FETCH_SIZE"])] = ("coag_sample", 1.0)
for k, (short, f) in names.items():
    if k in out and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        traffic[short] = {"bytes_per_launch": (out[k]["FETCH_SIZE"] * f + out[k]["WRITE_SIZE"]) * 1024.0,
                          "fetch_size_kib": out[k]["FETCH_SIZE"], "write_size_kib": out[k]["WRITE_SIZE"],
                          "fetch_factor": f}
if traffic:
    json.dump(traffic, open(os.path.join(root, f"pmc_{tag}_traffic.json"), "w"), indent=1)
