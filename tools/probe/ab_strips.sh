fmt='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d.get("k", d.get("p_zero")), d["code_lengths"], "dec", d["decode_GBps"], "body", d.get("dec_body_ms", d.get("phase_ms",{}).get("dec_body")), d["verified"])'
for sw in 1 0; do echo "== ET_NO_STRIPS=$sw"; ET_NO_STRIPS=$sw timeout -k 10 300 python3 tools/probe/flat_alphabets.py 268435456 3,5,6,7,10,12 2>/dev/null | python3 -c "$fmt"; ET_NO_STRIPS=$sw timeout -k 10 300 python3 tools/probe/sparse_streams.py 268435456 0.6,0.75,0.9,0.97 2>/dev/null | python3 -c "$fmt"; done
