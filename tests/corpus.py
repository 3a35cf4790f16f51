"""Deterministic synthetic inputs shared by tests, smoke() and bench.py.

None of the reference's benchmark corpora exist offline, so text-like streams are
i.i.d. order-0 samples of the byte distribution of res/a_midsummer_nights_dream.txt
(93 symbols, entropy ~4.69 bits/byte, no NUL, max code length 17).
"""
import os

import numpy as np

_RES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "res")


def midsummer():
    with open(os.path.join(_RES, "a_midsummer_nights_dream.txt"), "rb") as f:
        return f.read()


def midsummer_distribution():
    h = np.bincount(np.frombuffer(midsummer(), dtype=np.uint8), minlength=256).astype(np.float64)
    return h / h.sum()


def text_like(n, seed):
    """n bytes, order-0 samples of Midsummer's distribution (numpy Generator PCG64)."""
    p = midsummer_distribution()
    cdf = np.cumsum(p)
    cdf[-1] = 1.0
    rng = np.random.default_rng(seed)
    return np.searchsorted(cdf, rng.random(n), side="right").astype(np.uint8)


def tiled_midsummer(n):
    t = np.frombuffer(midsummer(), dtype=np.uint8)
    reps = (n + t.size - 1) // t.size
    return np.tile(t, reps)[:n].copy()


def uniform(n, seed, lo=0, hi=256):
    return np.random.default_rng(seed).integers(lo, hi, size=n, dtype=np.uint8 if hi <= 256 else np.uint16).astype(np.uint8)


def enwik_like_distribution():
    """A byte distribution shaped like a large mixed-markup corpus (the enwik configs of BASELINE.json, which do not
    exist offline): 206 symbols -- 96 "text" symbols, Zipf-like, carrying almost all of the mass, and 110 rare ones
    whose probabilities fall geometrically from 2^-12 to 2^-24.  Code lengths run from 3 to 24 bits or more: a
    12-bit lookup misses the long codes, which the text-like stream (93 symbols, 17 bits) never shows."""
    p = np.zeros(256, dtype=np.float64)
    order = np.random.default_rng(0xE9).permutation(256)  # which byte value gets which rank (fixed)
    common = 1.0 / np.arange(1, 97) ** 1.07
    rare = 2.0 ** -(12.0 + np.arange(110) / 9.0)
    p[order[:96]] = common / common.sum() * (1.0 - rare.sum())
    p[order[96:206]] = rare
    return p


def _sample(p, n, seed):
    cdf = np.cumsum(p)
    cdf[-1] = 1.0
    return np.searchsorted(cdf, np.random.default_rng(seed).random(n), side="right").astype(np.uint8)


def enwik_like(n, seed):
    return _sample(enwik_like_distribution(), n, seed)


def from_env(var, n=None):
    """A real corpus, if the environment names one (ET_CORPUS_SHAKESPEARE / ET_CORPUS_ENWIK8 / ET_CORPUS_ENWIK9:
    SURVEY.md 8d) and the file exists: its bytes (the first n), else None."""
    path = os.environ.get(var)
    if not path or not os.path.isfile(path):
        return None
    return np.fromfile(path, dtype=np.uint8, count=-1 if n is None else n)


def _sample_torch(p, n, seed, device):
    import torch

    cdf = torch.cumsum(torch.tensor(p, dtype=torch.float64, device=device), 0)
    cdf[-1] = 2.0
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 26
    for s in range(0, n, step):
        m = min(step, n - s)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)  # (float32 cannot resolve 2^-24)
        out[s : s + m] = torch.searchsorted(cdf, u, right=True).clamp_(max=255).to(torch.uint8)
    return out


def enwik_like_torch(n, seed, device):
    return _sample_torch(enwik_like_distribution(), n, seed, device)


def text_like_torch(n, seed, device):
    """Same distribution generated on the device (bench sizes): torch.multinomial-free
    inverse-CDF sampling with a seeded torch generator."""
    import torch

    p = torch.tensor(midsummer_distribution(), dtype=torch.float64, device=device)
    cdf = torch.cumsum(p, 0).to(torch.float32)
    cdf[-1] = 2.0
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 26
    for s in range(0, n, step):
        m = min(step, n - s)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float32)
        out[s : s + m] = torch.searchsorted(cdf, u, right=True).clamp_(max=255).to(torch.uint8)
    return out
