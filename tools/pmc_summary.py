#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/.../*_counter_collection.csv):
per et:: kernel, the mean of every counter over its dispatches."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "et::" not in k:
            continue
        name = k.split("(")[0].replace("void ", "").replace("et::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[k]["dispatches"] = len(next(iter(cs.values())))
json.dump(out, sys.stdout, indent=1)
print()
