"""Executable model (numpy, CPU) of the exact-parallel evaluation of a strict left-to-right f64 chain
acc_i = fl(acc_{i-1} + x_i), x_i >= 0 — the algorithm kernels_cdf_par.hip implements (DESIGN.md "Exact parallel CDF").

Idea: inside one binade [2^b, 2^(b+1)) every add rounds to the grid u = 2^(b-52); whether a tie rounds up depends only
on the parity of acc/u.  So a chunk of terms acts on acc as a 2-state transducer: acc_out = acc_in + D[parity(acc_in)],
and D0/D1 are obtained by running the TRUE f64 chain from the two reference starts 2^b and 2^b + u.  The transducer is
valid for the real acc_in iff acc_in is in binade b and acc_in + D stays below 2^(b+1); chunks where that cannot be
established ("dirty": a binade crossing, or the start 0) are evaluated by their true chain, in order.
"""
import math
import numpy as np


def serial_chain(x, acc0=0.0):
    acc = np.float64(acc0)
    out = np.empty(len(x), np.float64)
    for i, v in enumerate(x):
        acc = acc + np.float64(v)
        out[i] = acc
    return out


def _expo(v):
    return math.frexp(v)[1] - 1 if v > 0 else None      # floor(log2 v)


def parallel_chain(x, L):
    """returns (acc_in per chunk, final acc, n_dirty) using only chunk-local true chains + an ordered resolve"""
    n = len(x)
    T = (n + L - 1) // L
    chunks = [x[t * L:(t + 1) * L] for t in range(T)]
    # P1: approximate prefix at chunk boundaries (any summation order)
    csum = np.array([np.sum(c) for c in chunks])
    pend = np.cumsum(csum); pstart = pend - csum
    # binade guess + transducer by two reference chains
    guess, D0, D1 = [None] * T, [0.0] * T, [0.0] * T
    for t in range(T):
        b = _expo(float(pstart[t])) if pstart[t] > 0 else None
        if b is None or b < -1000:
            continue
        r0 = math.ldexp(1.0, b); u = math.ldexp(1.0, b - 52); r1 = r0 + u
        e0 = serial_chain(chunks[t], r0)[-1]; e1 = serial_chain(chunks[t], r1)[-1]
        top = math.ldexp(1.0, b + 1)
        if e0 < top and e1 < top:                       # reference chains stayed inside the binade
            guess[t], D0[t], D1[t] = b, e0 - r0, e1 - r1
    # ordered resolve
    acc = 0.0; acc_in = np.empty(T); dirty = 0
    for t in range(T):
        acc_in[t] = acc
        ok = False
        if guess[t] is not None and acc > 0 and _expo(acc) == guess[t]:
            u = math.ldexp(1.0, guess[t] - 52)
            par = int(acc / u) & 1                      # acc is a multiple of u inside its binade: exact
            cand = acc + (D1[t] if par else D0[t])
            if cand < math.ldexp(1.0, guess[t] + 1):
                acc = cand; ok = True
        if not ok:
            dirty += 1
            acc = float(serial_chain(chunks[t], acc)[-1])
    return acc_in, acc, dirty


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = 0
    for trial in range(300):
        n = int(rng.integers(1, 5000)); L = int(rng.choice([1, 3, 48, 64]))
        kind = trial % 6
        if kind == 0:
            x = rng.random(n)
        elif kind == 1:
            x = np.exp(rng.standard_normal(n) * 12).astype(np.float32).astype(np.float64)
        elif kind == 2:
            x = np.exp(rng.standard_normal(n) * 12); x = x / x.sum()
        elif kind == 3:
            x = np.ldexp(1.0, rng.integers(-80, 5, n))                 # powers of two: ties everywhere
        elif kind == 4:
            x = np.ldexp(rng.integers(1, 4, n).astype(np.float64), rng.integers(-60, -40, n)); x[rng.integers(0, n)] = 1.0
        else:
            x = np.sort(np.exp(rng.standard_normal(n) * 20))           # increasing: a crossing at almost every step
        if trial % 7 == 0:
            x[rng.integers(0, n, max(1, n // 10))] = 0.0
        ref = serial_chain(x)
        acc_in, total, dirty = parallel_chain(x, L)
        assert total == ref[-1], (trial, total, ref[-1])
        T = (n + L - 1) // L
        for t in range(1, T):
            assert acc_in[t] == ref[t * L - 1], (trial, t)
        worst = max(worst, dirty / T)
    print("exact-parallel chain model OK; worst dirty fraction", worst)
