"""The CDF kernels divide every term of a row by one of three row-wide divisors (sum of exps, norm, sum2; src/main.rs:799, 763-765, 779).
csrc/cz_cdf_common.h: div_by() replaces the IEEE division by a multiply and four fused multiply-adds on the correctly rounded reciprocal.
This is the exact rational model of that sequence against the division itself — quotients next to rounding midpoints, divisors whose
significand is (almost) all ones, f32-valued dividends like the kernel's exps.  The device function is checked in test_gpu_kernels.py."""
import math
import random
import struct
from fractions import Fraction as F


def rn(x):                       # Fraction -> nearest double, ties to even
    return float(x)


def fma(a, b, c):
    return rn(F(a) * F(b) + F(c))


def div_by(a, b, y):             # csrc/cz_cdf_common.h, statement by statement
    q = rn(F(a) * F(y))
    r = fma(-b, q, a)
    q = fma(r, y, q)
    r = fma(-b, q, a)
    return fma(r, y, q)


def operands(n, seed):
    rnd = random.Random(seed)
    for _ in range(n):
        u = rnd.random()
        if u < 0.5:
            b = rnd.uniform(1.0, 70000.0)
        elif u < 0.7:
            b = float(rnd.randint(1, 65536 + 256))
        elif u < 0.9:
            b = math.ldexp(1.0 + (2 ** 52 - 1 - rnd.randint(0, 3)) / 2 ** 52, rnd.randint(0, 16))      # significand (almost) all ones
        else:
            b = 1.0 + rnd.random() * 2.0 ** -rnd.randint(1, 40)                                          # norm, sum2: just above 1
        k = rnd.random()
        if k < 0.35:
            a = struct.unpack("f", struct.pack("f", math.ldexp(rnd.random(), -rnd.randint(0, 140))))[0]  # an f32 like exp(l - max)
        elif k < 0.8:
            q = math.ldexp(1 + rnd.getrandbits(52) / 2 ** 52, -rnd.randint(0, 60))                       # a / b next to a rounding midpoint
            a = rn(F(b) * (F(q) + F(math.ulp(q)) / 2))
            if rnd.random() < 0.5:
                a = math.nextafter(a, rnd.choice([0.0, math.inf]))
        elif k < 0.9:
            a = rnd.uniform(0.0, 1.0)
        else:
            a = rnd.choice([0.0, 1.0, 2.0 ** -149, 1.862645149230957e-09, b, math.nextafter(b, 0.0)])
        yield a, b


def test_five_operation_division_equals_ieee_division():
    n = 0
    for a, b in operands(40000, 20261004):
        assert div_by(a, b, 1.0 / b) == a / b, (a.hex(), b.hex())
        n += 1
    assert n == 40000
