#!/bin/bash
# A/B runs of bench.py under different environments, one line each.  Usage (through gpurun, from the repo root):
#   bash tools/ab_env.sh "name1:VAR=1 VAR2=x" "name2:VAR=2" ...      (name "base": no variables)
# ET_AB_ARGS: extra bench.py arguments (default: --steps 20 --warmup 3 --no-cpu-baseline --no-second-workload)
ARGS=${ET_AB_ARGS:---steps 20 --warmup 3 --no-cpu-baseline --no-second-workload}
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}; [ "$vars" = "$spec" ] && vars=""
  env $vars timeout -k 10 300 python bench.py $ARGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/ab_$name.err; exit 1; }
  python - "$name" <<'PY'
import json,sys
v=sys.argv[1]
d=json.load(open(f"gpurun_out/ab_{v}.json"))
p=d["phase_ms"]; w=d.get("workloads",{}).get("enwik-like",{}); q=w.get("phase_ms",{})
print(f"{v:12s} value {d.get('value', d.get('value_unverified', 0)):7.1f} cold {d.get('value_cold',0):7.1f} | hist {p['hist']:.4f} scan {p['enc_scan']:.4f} body {p['enc_body']:.4f} enc {p['enc_total']:.4f} | sync {p['dec_sync_first']:.4f} dbody {p['dec_body']:.4f} dec {p['dec_total']:.4f}" + (f" | enwik rt {w.get('round_trip_GBps',0):.1f} dec {q.get('dec_total',0):.4f} enc {q.get('enc_total',0):.4f}" if w else ""))
PY
done
