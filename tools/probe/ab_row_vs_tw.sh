# the row walk against the tree walk on codes of 7 and 8 bits with many short codewords (variants/libet_${VAR:-q8}.so: et::quick_to_synchronise lets L = 7 through)
KS=${KS:-136,150,160,170,180,190,200,210,220,230}
N=${N:-1073741824}
fmt='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["k"], d["code_lengths"], "dec", d["decode_GBps"], "sync", d["dec_sync_ms"], "body", d["dec_body_ms"], d["decode_path"][:18], d["verified"])'
echo "== row walk"; timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
echo "== tree walk where the estimate allows"; ET_LIB_PATH=$PWD/variants/libet_${VAR:-q8}.so timeout -k 10 500 python3 tools/probe/flat_alphabets.py $N $KS 2>/dev/null | python3 -c "$fmt"
