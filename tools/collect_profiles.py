#!/usr/bin/env python3
"""gpurun_out/ (tools/refresh_profiles.sh <tag>) -> profiles/<tag>_*: the bench line, the
rocprofv3 kernel-stats summary of the same command (et:: kernels first), the PMC summary."""
import csv, glob, json, shutil, subprocess, sys

tag = sys.argv[1]
shutil.copy(f"gpurun_out/{tag}_bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"gpurun_out/{tag}_bench_under_rocprof.json", f"profiles/{tag}_bench_under_rocprof.json")
src = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: ("et::" not in r["Name"], -float(r["TotalDurationNs"])))
with open(f"profiles/{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
subprocess.check_call([sys.executable, "tools/pmc_to_profile.py", tag])
d = json.load(open(f"profiles/{tag}_bench.json"))
print("value", d["value"], d["unit"], "| encode", d["encode_GBps"], "decode", d["decode_GBps"], "| roofline", d["roofline"])
for r in rows[:8]:
    print(f"  {r['Name'].split('(')[0][:50]:52s} avg {float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']}")
