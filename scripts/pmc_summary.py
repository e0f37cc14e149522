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
