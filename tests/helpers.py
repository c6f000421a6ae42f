"""Shared helpers for the parity tests."""
import numpy as np

MODES = {"canonical": 0, "forward": 1, "reverse": 2}
CODE = {"A": 0, "C": 1, "T": 2, "G": 3}


def kmer_string_to_int(s):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ch]
    return v


def expected_arrays(case):
    """golden case -> (keys python-int list, counts list)"""
    keys = [kmer_string_to_int(s) for s, _ in case["expected"]]
    counts = [c for _, c in case["expected"]]
    return keys, counts


def join128(hi, lo):
    return [(int(h) << 64) | int(l) for h, l in zip(hi, lo)]


def python_count(bases, k, mode=0):
    """Third, independent statement of the semantics in pure Python (small inputs
    only): used to cross-check the C oracle itself."""
    if isinstance(bases, bytes):
        bases = bases.decode("ascii")
    comp = {0: 2, 1: 3, 2: 0, 3: 1}
    out = {}
    run = []
    for ch in bases:
        c = CODE.get(ch.upper()) if ch.upper() in CODE and ch in "ACGTacgt" else None
        if c is None:
            run = []
            continue
        run.append(c)
        if len(run) > k:
            run.pop(0)
        if len(run) == k:
            f = 0
            for x in run:
                f = (f << 2) | x
            r = 0
            for x in reversed(run):
                r = (r << 2) | comp[x]
            m = f if mode == 1 else r if mode == 2 else min(f, r)
            out[m] = out.get(m, 0) + 1
    keys = sorted(out)
    return keys, [out[x] & 0xFFFFFFFF for x in keys]


def random_reads(rng, n_reads, min_len, max_len, n_rate=0.01, lower_rate=0.05):
    alpha = np.array(list("ACGT"))
    parts = []
    for _ in range(n_reads):
        ln = int(rng.integers(min_len, max_len + 1))
        s = alpha[rng.integers(0, 4, ln)]
        s = np.where(rng.random(ln) < n_rate, "N", s)
        low = rng.random(ln) < lower_rate
        s = np.where(low, np.char.lower(s), s)
        parts.append("".join(s))
        parts.append(".")
    return "".join(parts)
