#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel.
usage: tools/pmc_summary.py <counter_collection.csv> [out.csv]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("Grid_Size"), r.get("Workgroup_Size"))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(["kernel", "counter", "mean", "launches", "vgpr", "sgpr", "grid", "workgroup"])
for k in sorted(agg):
    if "szg::" not in k:
        continue
    for c in sorted(agg[k]):
        v = agg[k][c]
        w.writerow([k, c, f"{sum(v) / len(v):.6g}", len(v), *meta[k]])
