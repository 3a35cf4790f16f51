#!/usr/bin/env python3
"""What the compiler-placement pins hold in place, checked in the gfx950 ISA.

Round 3's largest gains were placements: where `s_waitcnt vmcnt` sits relative to a prefetch (K4, D3), and 128 compares that
must NOT be hoisted in front of every block (D1).  None of them changes a result, so the parity tests stay green when a new
hipcc or an innocent edit undoes one; only the bench would move.  This module compiles the two kernel files to assembly
(`hipcc -S --cuda-device-only`, no GPU needed) and checks the structure itself:

  K4  k_encode_tiles<4096>   no `s_waitcnt vmcnt` between the round loop's two barriers; the wait for the NEXT round's chunk
                             stands directly behind the second barrier, in front of the round's stores
  D3  k_dec_write_wave<8>    no `s_waitcnt vmcnt` between the unit loop's head and the next unit's loads (a unit does not wait
                             for its own output stores before it asks for anything); none inside the walk; the wait for the
                             next unit's words stands in front of the last window's stores
  D1  k_tw_sync              no SGPR spills (the hoisted edge compares cost 251), no v_writelane, <= 80 VGPRs
  occupancy                  VGPR counts / static LDS that give K1 4 x 512 threads, K4 8 and D1 / D3 3 x 8 wavefronts per SIMD/CU

    python tools/isa_guard.py            # build + check, prints one line per check
    python tools/isa_guard.py --drop     # the same with -DET_GUARD_DROP_PINS: the checks that the pins hold must FAIL
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ("et_kernels", "et_treewalk")


def compile_isa(out_dir, extra_flags=()):
    """hipcc -S of both kernel files into out_dir (in parallel); returns {source: path}."""
    procs, paths = [], {}
    for src in SOURCES:
        paths[src] = os.path.join(out_dir, src + ".s")
        cmd = [HIPCC, "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "entreepy_amd", "csrc"), "--offload-arch=gfx950", "-S",
               "--cuda-device-only", *extra_flags, os.path.join(ROOT, "entreepy_amd", "csrc", src + ".hip"), "-o", paths[src]]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    for src, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc -S {src}.hip failed:\n{err.decode()[-2000:]}")
    return paths


def function_lines(asm_path, mangled_part):
    """The instructions and labels of one kernel (stripped lines, comments kept out except loop notes)."""
    lines = open(asm_path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(mangled_part) + r"\w*:", l))
    b = next(i for i in range(a, len(lines)) if lines[i].strip() == "s_endpgm")
    return [l.strip() for l in lines[a : b + 1]]


def metadata(asm_path):
    """{kernel name: {vgpr_count, sgpr_count, sgpr_spill_count, vgpr_spill_count, lds}} from the amdhsa.kernels notes."""
    s = open(asm_path).read()
    md = s[s.index("amdhsa.kernels:") :]
    out = {}
    for blk in md.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))  # noqa: E731
        out[name] = {"vgpr_count": g("vgpr_count"), "sgpr_count": g("sgpr_count"), "sgpr_spill_count": g("sgpr_spill_count"),
                     "vgpr_spill_count": g("vgpr_spill_count"), "lds": g("group_segment_fixed_size")}
    return out


def _is_vm_wait(l):
    return l.startswith("s_waitcnt") and "vmcnt(" in l


def check_k4(lines):
    """-> list of (name, ok, detail)."""
    bar = [i for i, l in enumerate(lines) if l == "s_barrier"]
    # the round loop's barrier pair: the two consecutive barriers with the most `ds_or_b32` (the append steps) between them
    pairs = [(sum(1 for l in lines[a:b] if l.startswith("ds_or_b32")), a, b) for a, b in zip(bar, bar[1:])]
    n_or, b1, b2 = max(pairs)
    out = [("K4: the round loop's append steps lie between two barriers", n_or >= 8, f"{n_or} ds_or_b32 between the barriers")]
    # the round begins where the next chunk is asked for: the last 16-byte load in front of the first barrier
    loads = [i for i in range(b1) if lines[i].startswith("global_load_dwordx4")]
    top = loads[-1] if loads else b1
    waits = [i for i in range(top, b2) if _is_vm_wait(lines[i])]
    out.append(("K4: no s_waitcnt vmcnt between the round's prefetch and its second barrier", not waits, f"{len(waits)} waits in {b2 - top} lines"))
    nxt = next((i for i in range(b2 + 1, len(lines)) if lines[i] and not lines[i].startswith((";", "."))), None)
    out.append(("K4: the next chunk is waited for directly behind the append barrier", nxt is not None and _is_vm_wait(lines[nxt]), lines[nxt] if nxt else "-"))
    store = next((i for i in range(b2, len(lines)) if lines[i].startswith("global_store")), None)
    out.append(("K4: ... in front of the round's stores", store is not None and nxt is not None and nxt < store, f"store {store}, wait {nxt}"))
    return out


def check_d3(lines):
    heads = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
    # the unit loop: the depth-1 loop that holds the walk (the most ds_read_b64 behind its head)
    head = max(heads, key=lambda h: sum(1 for l in lines[h:] if l.startswith("ds_read_b64")))
    first_load = next(i for i in range(head, len(lines)) if lines[i].startswith("global_load"))
    waits = [i for i in range(head, first_load) if _is_vm_wait(lines[i])]
    out = [("D3: a unit asks for the next unit's words without waiting for its own stores (no vmcnt wait at the unit loop's head)", not waits,
            f"{len(waits)} waits in the {first_load - head} lines between the loop's head and its first load")]
    reads = [i for i in range(head, len(lines)) if lines[i].startswith("ds_read_b64")]
    walk = reads[: min(10, len(reads))]  # the one-window walk: nine word loops and the tail
    inside = [i for i in range(walk[0], walk[-1]) if _is_vm_wait(lines[i])]
    out.append(("D3: no s_waitcnt vmcnt inside the walk", not inside, f"{len(inside)} waits between the walk's first and tenth lookup"))
    stores = [i for i, l in enumerate(lines) if l.startswith("global_store_dwordx4") and l.endswith(" nt")]
    last = stores[-1]
    prev_store = max([i for i, l in enumerate(lines[:last]) if l.startswith("global_store")] or [head])
    take = [i for i in range(prev_store, last) if lines[i].startswith("s_waitcnt") and "vmcnt(0)" in lines[i]]
    out.append(("D3: the next unit's words are taken in front of the last window's stores", len(take) == 1, f"{len(take)} vmcnt(0) between the stores at {prev_store} and {last}"))
    return out


def check_d1(lines, md):
    m = next(v for k, v in md.items() if "k_tw_sync" in k)
    wl = sum(1 for l in lines if l.startswith("v_writelane"))
    return [("D1: k_tw_sync spills no SGPRs (the edge blocks' step-limit compares stay inside the edge branch)", m["sgpr_spill_count"] == 0 and wl == 0,
             f"sgpr_spill_count {m['sgpr_spill_count']}, {wl} v_writelane"),
            ("D1: k_tw_sync <= 80 VGPRs, nothing in scratch (6 wavefronts per SIMD)", m["vgpr_count"] <= 80 and m["vgpr_spill_count"] == 0, f"{m['vgpr_count']} VGPRs, {m['vgpr_spill_count']} spilled")]


def check_occupancy(md):
    def of(part):
        return next(v for k, v in md.items() if part in k)

    k1, k4, d3 = of("k_hist_tiles"), of("k_encode_tilesILj4096"), of("k_dec_write_waveILi8ELb0")
    return [("K1: 32 KiB of counters, <= 64 VGPRs (4 workgroups of 512 threads per CU)", k1["lds"] == 32768 and k1["vgpr_count"] <= 64, str(k1)),
            ("K4: <= 64 VGPRs and an 18 KiB ring + table (8 workgroups per CU)", k4["vgpr_count"] <= 64 and k4["lds"] <= 20480 and k4["sgpr_spill_count"] == 0, str(k4)),
            ("D3: <= 64 VGPRs, nothing spilled (3 workgroups of 8 wavefronts per CU beside its tables)", d3["vgpr_count"] <= 64 and d3["vgpr_spill_count"] == 0 and d3["sgpr_spill_count"] == 0, str(d3))]


def run_checks(paths):
    md = {}
    for p in paths.values():
        md.update(metadata(p))
    res = []
    res += check_k4(function_lines(paths["et_kernels"], "k_encode_tilesILj4096"))
    res += check_d3(function_lines(paths["et_kernels"], "k_dec_write_waveILi8ELb0"))
    res += check_d1(function_lines(paths["et_treewalk"], "k_tw_sync"), md)
    res += check_occupancy(md)
    return res


def main():
    drop = "--drop" in sys.argv
    with tempfile.TemporaryDirectory() as d:
        res = run_checks(compile_isa(d, ("-DET_GUARD_DROP_PINS",) if drop else ()))
    for name, ok, detail in res:
        print(("ok    " if ok else "FAILED"), name, "--", detail)
    sys.exit(0 if all(ok for _, ok, _ in res) else 1)


if __name__ == "__main__":
    main()
