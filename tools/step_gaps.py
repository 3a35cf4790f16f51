"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV: start, duration and the
idle gap in front of every kernel (python tools/step_gaps.py <..._kernel_trace.csv>)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48]) for r in rows)
idx = [i for i, e in enumerate(ev) if 'k_hist_tiles' in e[2]]
a, b = idx[-3], idx[-2]
t0, busy_end, tot_gap = ev[a][0], None, 0
for s, e, name in ev[a:b + 1]:
    gap = (s - busy_end) / 1000 if busy_end else 0
    if busy_end and s > busy_end:
        tot_gap += s - busy_end
    print(f"{(s - t0) / 1000:9.1f} us  dur {(e - s) / 1000:8.1f}  gap {gap:7.1f}  {name}")
    busy_end = max(busy_end or e, e)
print("step span", (ev[b][0] - ev[a][0]) / 1000, "us; idle gaps total", tot_gap / 1000)
