# Codes of L and L + 1 bits (k symbols of equal weight): the decode's choice (et::quick_to_synchronise) against "never try the tree walk"
# (ET_NO_QUICK_SYNC=1: rounds 1-3) and "always try it" (ET_QUICK_SYNC_ALWAYS=1: the first sweep's own verdict decides) -- where the
# thresholds in et_rowsync_host.cpp come from.  Through gpurun:  KS=31,33,34 bash tools/probe/ab_flat_rule.sh
KS=${KS:-3,5,6,7,10,12,14,17,20,26,30,31,33,34,36,40,50,58,60,61,62,65,68,72,80,100,110,112,116,120,124}
N=${N:-268435456}
fmt='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["k"], d["code_lengths"], "dec", d["decode_GBps"], "sync", d["dec_sync_ms"], "body", d["dec_body_ms"], d["decode_path"][:18], d["verified"])'
echo "== never (ET_NO_QUICK_SYNC=1)"; ET_NO_QUICK_SYNC=1 timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
echo "== always (ET_QUICK_SYNC_ALWAYS=1)"; ET_QUICK_SYNC_ALWAYS=1 timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
echo "== the rule"; timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
