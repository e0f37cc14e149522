"""HBM traffic per launch of the pair sampler FOR ONE RUN SHAPE (bench.py --warmup W --steps K):
reads the FETCH_SIZE and WRITE_SIZE passes of scripts/gpu_pmc_shape.sh and the bench line each pass
printed, averages the counters over the sampler's LAST K dispatches (the timed steps; --resort-steps 0,
so nothing follows them) and records the events and candidates per launch of those same steps beside
them.  Merges the entry into profiles/pmc_traffic.json under "shapes" (key "w{W}s{K}").

FETCH_SIZE / WRITE_SIZE are in KiB.  The sampler's gathers (4 to 32 bytes per lane at independent
addresses) are calibrated by scripts/gather_probe.hip: FETCH_SIZE counts 64 B per such load, used
raw (MI355X_MICROARCH.md: the gfx950 halving applies to coalesced streams, not to these).

usage: python3 scripts/pmc_shape.py <gpurun_out> <tag> <warmup> <steps> [out.json]"""
import csv, glob, json, os, sys

root, tag, W, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
out_path = sys.argv[5] if len(sys.argv) > 5 else os.path.join(root, f"pmc_{tag}_shape.json")
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(root, f"pmcs_{tag}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if row["Counter_Name"] == counter and ("k_coag_lane<" in k or "k_coag_sample<" in k):
                rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), k.split("(")[0].replace("void ", "")))
    rows.sort()
    if len(rows) < K:
        sys.exit(f"{counter}: {len(rows)} sampler dispatches, expected at least {K}")
    timed = rows[-K:]
    res[counter] = sum(v for _, v, _ in timed) / K
    res["kernel"] = timed[-1][2]
    res["dispatches_seen"] = len(rows)
    line = None
    for ln in open(os.path.join(root, f"pmcs_{tag}_{counter}.log")):
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    if line is None:
        sys.exit(f"no bench line in pmcs_{tag}_{counter}.log")
    res.setdefault("events_per_launch", line["events_per_step"])
    res.setdefault("candidates_per_launch", line["candidate_pairs_per_sec"] * line["ms_per_step"] * 1e-3)
    res.setdefault("workload", line["config"]["workload"])
    if abs(line["events_per_step"] - res["events_per_launch"]) > 0:
        sys.exit("the two passes did not run the same steps")
entry = {"kernel": res["kernel"], "warmup": W, "steps": K,
         "fetch_size_kib": res["FETCH_SIZE"], "write_size_kib": res["WRITE_SIZE"],
         "bytes_per_launch": (res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0,
         "events_per_launch": res["events_per_launch"], "candidates_per_launch": res["candidates_per_launch"],
         "bytes_per_event": (res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0 / res["events_per_launch"],
         "workload": res["workload"], "averaged_over": f"the sampler's last {K} of {res['dispatches_seen']} dispatches"}
doc = {}
if os.path.exists(out_path):
    doc = json.load(open(out_path))
doc.setdefault("shapes", {})[f"w{W}s{K}"] = entry
json.dump(doc, open(out_path, "w"), indent=1)
print(f"w{W}s{K}", json.dumps(entry))
