# Codes of 7 and 8 bits with many short codewords: the row walk (ET_NO_QUICK_SYNC=1) against the tree walk (ET_QUICK_SYNC_ALWAYS=1; the
# row walk is where it ends up when its blocks give up) and the rule's choice.  Through gpurun:  bash tools/probe/ab_row_vs_tw.sh
KS=${KS:-130,136,140,150,160,180,200,205,210,220,230,240,250}
N=${N:-1073741824}
fmt='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["k"], d["code_lengths"], "dec", d["decode_GBps"], "sync", d["dec_sync_ms"], "body", d["dec_body_ms"], d["decode_path"][:18], d["verified"])'
echo "== row walk (ET_NO_QUICK_SYNC=1)"; ET_NO_QUICK_SYNC=1 timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
echo "== tree walk first (ET_QUICK_SYNC_ALWAYS=1)"; ET_QUICK_SYNC_ALWAYS=1 timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
echo "== the rule"; timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
