#!/bin/bash
# Build library variants with different -D knobs (HERE, hipcc cross-compiles) and bench
# each on the GPU box:  tools/ab_variants.sh build "name:-DX=1 -DY=2" ... ; then
# gpurun -- 'bash tools/ab_variants.sh run name ...'
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p build/variants
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude --offload-arch=gfx950 $flags -shared -o build/variants/lib_$name.so \
      entreepy_amd/csrc/et_kernels.hip entreepy_amd/csrc/et_api.cpp entreepy_amd/csrc/et_codebook.cpp entreepy_amd/csrc/et_io.cpp entreepy_amd/csrc/et_tables.cpp -lpthread 2>&1 | grep -E "error" || true
    echo "built $name ($flags)"
  done
else
  for name in "$@"; do
    for rep in 1 2; do
      ET_LIB_PATH=$PWD/build/variants/lib_$name.so timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/ab_$name.err; continue; }
      python - "$name" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json")); p=d["phase_ms"]
print(f"{sys.argv[1]:28s} value {d['value']:7.1f} enc {d['encode_GBps']:7.1f} dec {d['decode_GBps']:6.1f} | hist {p['hist']:.3f} body {p['enc_body']:.3f} | sync {p['dec_sync']:.3f} write {p['dec_body']:.3f}")
PY
    done
  done
fi
