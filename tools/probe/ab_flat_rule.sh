KS=${KS:-3,5,6,7,10,12,14,17,20,26,31,40,50,100}
echo "== rule as is"; timeout -k 10 500 python3 tools/probe/flat_alphabets.py 268435456 $KS 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['k'], d['code_lengths'], 'dec', d['decode_GBps'], 'sync', d['dec_sync_ms'], 'body', d['dec_body_ms'], d['decode_path'][:18], d['verified'])"
echo "== no rule"; ET_PROBE_NO_EXH_RULE=1 ET_LIB_PATH=$PWD/variants/libet_noexh.so timeout -k 10 500 python3 tools/probe/flat_alphabets.py 268435456 $KS 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['k'], d['code_lengths'], 'dec', d['decode_GBps'], 'sync', d['dec_sync_ms'], 'body', d['dec_body_ms'], d['decode_path'][:18], d['verified'])"
