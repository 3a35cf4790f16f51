#!/usr/bin/env python3
"""Basic blocks of one kernel in a gfx950 ISA dump, with VALU / SALU / LDS / memory instruction counts, branch targets and
every s_waitcnt vmcnt -- how round 3 found the edge compares hoisted in front of every block of k_tw_sync (337 VALU in a loop
preheader) and the vmcnt(0) that made K4's prefetch wait for itself.
  hipcc -O3 -std=c++17 -Iinclude -Ientreepy_amd/csrc --offload-arch=gfx950 -S --cuda-device-only entreepy_amd/csrc/et_kernels.hip -o /tmp/k.s
  python tools/isa_blocks.py /tmp/k.s k_encode_tilesILj4096 [min instructions per block]"""
import re
import sys

src, kernel = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kernel) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
blocks, cur = [], ["entry", [], start]
for i in range(start + 1, end + 1):
    l = lines[i]
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), [], i]
    elif t.startswith("; %bb."):
        blocks.append(cur)
        cur = [t.split()[1].rstrip(":"), [], i]
    elif t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t)
blocks.append(cur)
for name, ins, ln in blocks:
    if len(ins) < min_n:
        continue
    n = lambda *p: sum(1 for x in ins if x.startswith(p))
    br = " ".join(x.split()[0][2:] + ">" + x.split()[1] for x in ins if x.startswith(("s_cbranch", "s_branch")))
    waits = " ".join(re.findall(r"vmcnt\(\d+\)", " ".join(x for x in ins if x.startswith("s_waitcnt"))))
    print(f"{name:10s} L{ln - start:5d} n={len(ins):4d} valu={n('v_'):4d} salu={n('s_'):4d} lds={n('ds_'):3d} mem={n('global_', 'buffer_', 'scratch_'):3d}  {br}  {waits}")
