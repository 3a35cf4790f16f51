#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (et_encode / et_decode):
pageable host memory in, pageable host memory out.  Never the headline `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import entreepy_amd as E  # noqa: E402
from tests import corpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
data = corpus.text_like(n, 1)
ctx = E.Context(0)
ctx.reserve(n)
et = ctx.encode(data)
for label, fn, arg in (("et_encode", ctx.encode, data), ("et_decode", ctx.decode, et[4:])):
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        out = fn(arg)
        ts.append(time.perf_counter() - t)
    print(f"{label}: {n / min(ts) / 1e9:.2f} GB/s of text, host to host, n = {n} B (best of 4, includes H2D + D2H + numpy copy-out)")
assert ctx.decode(et[4:]) == data.tobytes()
