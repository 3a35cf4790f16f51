"""Test infrastructure: the product's group sequence over a CPU stand-in for the GPU (see shard_cpu.cpp)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def shard_cpu_lib():
    """libet_shard_cpu.so (built with g++ on first use): csrc/et_shard_seq.cpp + shard_cpu.cpp + the oracle, with the
    group entry points declared as the product's are."""
    global _lib
    if _lib is None:
        from entreepy_amd import _native as N

        subprocess.check_call(["make", "-s", "-C", _HERE, "libet_shard_cpu.so"])
        _lib = N.declare(ctypes.CDLL(os.path.join(_HERE, "libet_shard_cpu.so")), N.GROUP_SYMBOLS)
    return _lib
