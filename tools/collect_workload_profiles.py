#!/usr/bin/env python3
"""gpurun_out/ (tools/profile_workloads.sh <tag>) -> profiles/<tag>_<workload>_kernel_stats.csv (et:: kernels first) and
profiles/<tag>_<workload>.json (the workload's line from tools/run_workload.py, taken under rocprofv3)."""
import csv, glob, json, shutil, sys

tag = sys.argv[1]
for src in sorted(glob.glob(f"gpurun_out/prof_{tag}_*")):
    w = src.split(f"prof_{tag}_")[1]
    stats = glob.glob(f"{src}/**/*kernel_stats.csv", recursive=True)
    if not stats:
        continue
    rows = list(csv.DictReader(open(stats[0])))
    rows.sort(key=lambda r: ("et::" not in r["Name"], -float(r["TotalDurationNs"])))
    short = w.replace("-", "")
    with open(f"profiles/{tag}_{short}_kernel_stats.csv", "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    try:
        shutil.copy(f"gpurun_out/{tag}_{w}.json", f"profiles/{tag}_{short}_under_rocprof.json")
    except OSError:
        pass
    print(w)
    for r in rows[:6]:
        print(f"  {r['Name'].split('(')[0][:50]:52s} avg {float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']}")
