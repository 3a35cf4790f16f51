# A/B of k_row_sync variants on the 4 GiB uniform stream (through gpurun): variants/libet_<name>.so from tools/build_variant.sh, e.g. ch8 = -DET_ROW_CHUNK_BLOCKS=8
for spec in ${ET_AB_SPECS:-base ch8 ch2 rs8 base2}; do
  if [ "$spec" = base ] || [ "$spec" = base2 ]; then lib=""; else lib="ET_LIB_PATH=$PWD/variants/libet_$spec.so"; fi
  env $lib python tools/run_workload.py uniform255-4G 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())['uniform255-4G']; p=d['phase_ms']
print('$spec', 'sync', p['dec_sync'], 'write', p['dec_body'], 'dec', p['dec_total'], 'GB/s', d['decode_GBps'], d['verified'])"
done
