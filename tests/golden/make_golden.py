"""Regenerates tests/golden/*.et and golden.json from the oracle.

The reference (Zig) cannot be built or run in this image, so these vectors come from
oracle/et_oracle.c -- itself pinned to the reference's one exact known answer
(README.md:51: nice.shakespeare.txt 477 B -> 374 B) and to the vectors an independent
model of the source produced during the survey (42-byte hex for
test.txt, three SHA-256s).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402

CASES = {
    # name: bytes
    "aaaa": b"aaaa",                       # single symbol -> header only (Q2)
    "abab": b"ab" * 10,
    "nul_mix": bytes([0, 1, 0, 2, 0, 0, 3, 1, 0, 2] * 20),  # NUL symbols (Q6)
    "all256": bytes(range(256)) * 3 + bytes(range(0, 256, 2)),  # 256 distinct (Q1)
    "all255": bytes(range(1, 256)) * 2 + b"\x07\x07\x09",
}


def main():
    manifest = {}
    for name in ("test.txt", "nice.shakespeare.txt", "a_midsummer_nights_dream.txt"):
        with open(os.path.join(HERE, "res", name), "rb") as f:
            text = f.read()
        et = O.encode(text)
        with open(os.path.join(HERE, name + ".et"), "wb") as f:
            f.write(et)
        manifest[name] = {"input": "res/" + name, "n": len(text), "et_len": len(et), "sha256": hashlib.sha256(et).hexdigest()}
    for name, text in CASES.items():
        et = O.encode(text)
        manifest[name] = {"input_hex": text.hex(), "n": len(text), "et_hex": et.hex(), "sha256": hashlib.sha256(et).hexdigest()}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "vectors")


if __name__ == "__main__":
    main()
