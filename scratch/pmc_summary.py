#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import csv, sys, glob, collections, json, os
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if not (k.startswith("void k_") or k.startswith("k_")): continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
