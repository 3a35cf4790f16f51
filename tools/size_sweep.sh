#!/bin/bash
# Round trip / encode / decode GB/s of bench.py at 16 MiB .. 3.75 GiB per step.  On the GPU box: bash tools/size_sweep.sh
for b in 16777216 67108864 268435456 1073741824 4026531840; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-workload --bytes $b > gpurun_out/sw_$b.json 2> gpurun_out/sw_$b.err || { echo "$b FAILED"; tail -3 gpurun_out/sw_$b.err; exit 1; }
  python - $b <<'PY'
import json,sys
b=sys.argv[1]
d=json.load(open(f"gpurun_out/sw_{b}.json"))
print(f"{int(b)>>20:6d} MiB  value {d['value']:7.1f}  enc {d['encode_GBps']:7.1f}  dec {d['decode_GBps']:7.1f}  ms/step {d['ms_per_step']:.4f}")
PY
done
