"""Adversarial columns for the tools/fuzz_*.py tools."""
import numpy as np

NULL = -(2**63)


def col_i64(rng, n):
    kind = rng.integers(0, 6)
    if kind == 0:
        a = rng.integers(-5, 5, n)
    elif kind == 1:
        a = rng.integers(-(2**62), 2**62, n)
    elif kind == 2:
        a = rng.choice(np.array([NULL, NULL + 1, 2**63 - 1, 0, -1, 1], np.int64), n)
    elif kind == 3:
        a = np.zeros(n, np.int64)
    elif kind == 4:
        a = np.full(n, NULL, np.int64)
    else:
        a = rng.integers(0, 1_000_000, n)
    a = a.astype(np.int64)
    if kind not in (3, 4) and rng.random() < 0.5:
        a[rng.random(n) < rng.choice([0.01, 0.5])] = NULL
    return a


def col_f64(rng, n):
    kind = rng.integers(0, 5)
    if kind == 0:
        v = rng.random(n) - 0.5
    elif kind == 1:
        v = rng.choice(np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.5, -1.5, 1e308, -1e308, 5e-324]), n)
    elif kind == 2:
        v = np.zeros(n)
    elif kind == 3:
        v = -np.zeros(n)
    else:
        v = rng.integers(-3, 3, n).astype(np.float64)
    if kind != 1 and rng.random() < 0.4:
        v[rng.random(n) < rng.choice([0.01, 0.5])] = np.nan
    return v.astype(np.float64)
