"""The observation-noise draw's tables (city2ba_amd/csrc/noise_tables.inc, r05): the committed file is what
tools/gen_noise_tables.py writes, its entries are the correctly rounded values they claim to be, and the table ALGORITHM
(camera_math.hpp: sincos_turns16, m2log_tab -- restated here in exact rational arithmetic on the tabled doubles, so that
the check does not depend on this machine's libm or on fused multiply-adds) stays within 1e-15 of the true functions."""
import math
import os
import re
import subprocess
import sys
from fractions import Fraction

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "city2ba_amd", "csrc", "noise_tables.inc")


def _table():
    rows = re.findall(r"^\{(\S+), (\S+)\},$", open(INC).read(), flags=re.M)
    return np.array([[float.fromhex(a), float.fromhex(b)] for a, b in rows])


def test_committed_file_is_what_the_generator_writes():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_noise_tables.py"), "--check"]).returncode == 0


def test_entries_are_correctly_rounded():
    t = _table()
    assert t.shape == (704, 2)
    mp.mp.dps = 50
    for h in (0, 1, 63, 64, 65, 128, 200, 255):
        assert t[h, 0] == float(mp.cospi(mp.mpf(h) / 128)) and t[h, 1] == float(mp.sinpi(mp.mpf(h) / 128))
    for l in (0, 1, 2, 127, 255):
        assert t[256 + l, 0] == float(mp.cospi(mp.mpf(l) / 32768)) and t[256 + l, 1] == float(mp.sinpi(mp.mpf(l) / 32768))
    # numpy's libm over all of them (its argument 2 pi h / 256 is itself rounded: 1e-15, not an ulp)
    a = 2 * np.pi * np.arange(256)
    assert np.max(np.abs(t[:256, 0] - np.cos(a / 256))) < 1e-15 and np.max(np.abs(t[256:512, 1] - np.sin(a / 65536))) < 1e-15
    assert t[0].tolist() == [1.0, 0.0] and t[64].tolist() == [0.0, 1.0] and t[128].tolist() == [-1.0, 0.0] and t[192].tolist() == [0.0, -1.0]
    assert t[256].tolist() == [1.0, 0.0]
    # the logarithm's intervals: the two that touch z = 1 have c = 1 exactly; every other invc is within 2^-8 of its centre
    assert t[512 + 79].tolist() == [1.0, 0.0] and t[512 + 80].tolist() == [1.0, 0.0]
    for i in (0, 17, 78, 81, 127):
        invc = t[512 + i, 0]
        assert t[512 + i, 1] == float(-2 * mp.log(1 / mp.mpf(invc)))


def _z_of(u):
    """camera_math.hpp: m2log_tab's split u = z 2^k, z in [0.6875, 1.375), and the interval of z"""
    hi = np.float64(u).view(np.uint64) >> np.uint64(32)
    tmp = (int(hi) - 0x3FE60000) & 0xFFFFFFFF
    i = (tmp >> 13) & 127
    k = tmp >> 20
    k = k - 4096 if k >= 2048 else k                      # arithmetic shift of the signed word
    return math.ldexp(u, -k), k, i


def test_table_algorithm_against_exact_arithmetic():
    t = _table()
    mp.mp.dps = 40
    rng = np.random.default_rng(5)
    # directions: the angle sum of the two tabled pairs, evaluated exactly, against the true cosine / sine
    for a in list(rng.integers(0, 65536, 300)) + [0, 1, 255, 256, 16383, 16384, 16385, 32768, 49152, 65535]:
        A, B = t[a >> 8], t[256 + (a & 255)]
        c = Fraction(A[0]) * Fraction(B[0]) - Fraction(A[1]) * Fraction(B[1])
        s = Fraction(A[1]) * Fraction(B[0]) + Fraction(A[0]) * Fraction(B[1])
        th = 2 * mp.pi * int(a) / 65536
        assert abs(mp.mpf(c.numerator) / c.denominator - mp.cos(th)) < 2.3e-16
        assert abs(mp.mpf(s.numerator) / s.denominator - mp.sin(th)) < 2.3e-16
    # -2 ln u for u = (w + 1) 2^-32: the series in r = z / c - 1 with exact r, against the true logarithm (relative)
    ws = list(rng.integers(0, 2 ** 32, 300)) + [0, 1, 2, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 3, 2 ** 32 - 2]
    for w in ws:
        u = (int(w) + 1) * 2.0 ** -32
        z, k, i = _z_of(u)
        assert 0.6875 <= z < 1.375 and 0 <= i < 128
        invc, m2logc = t[512 + i]
        r = Fraction(z) * Fraction(invc) - 1
        assert abs(r) < Fraction(1, 120)
        p = Fraction(-2, 7)
        for cf in (Fraction(1, 3), Fraction(-2, 5), Fraction(1, 2), Fraction(-2, 3), Fraction(1)):
            p = p * r + cf
        series = -2 * r + r * r * p
        got = mp.mpf(series.numerator) / series.denominator + mp.mpf(float(m2logc)) - 2 * k * mp.log(2)
        want = -2 * mp.log(mp.mpf(u))
        assert abs(got - want) <= 4e-16 * abs(want), (w, float(got), float(want))
    z, k, i = _z_of(1.0)                                   # w = 2^32 - 1: u = 1, the draw's magnitude is exactly zero
    assert (z, k, i) == (1.0, 0, 80)


def test_exponential_table_and_series():
    """entries [640, 704): 2^(j / 128) as 128 consecutive doubles, correctly rounded; camera_math.hpp: exp_tab's reduction and
    degree-5 series restated in exact arithmetic stay within 2e-16 (relative) of exp(x)"""
    t = _table()[640:].reshape(-1)
    assert t.shape == (128,)
    mp.mp.dps = 40
    for j in (0, 1, 2, 63, 64, 127):
        assert t[j] == float(mp.power(2, mp.mpf(j) / 128))
    assert t[0] == 1.0 and np.all(np.diff(t) > 0)
    hi, lo, inv = float.fromhex("0x1.62e42fee00000p-8"), float.fromhex("0x1.a39ef35793c76p-40"), float.fromhex("0x1.71547652b82fep+7")
    assert abs(mp.mpf(hi) + mp.mpf(lo) - mp.log(2) / 128) < mp.mpf(2) ** -90 and Fraction(hi) * (1 << 17) * (1 << 40) % 1 == 0
    rng = np.random.default_rng(9)
    for x in list(rng.uniform(-600, 600, 200)) + [0.0, 1e-9, -1e-9, 599.9, -599.9]:
        kf = round(float(x) * inv)
        r = Fraction(float(x)) - kf * Fraction(hi) - kf * Fraction(lo)
        assert abs(r) < Fraction(28, 10000)
        q = Fraction(1, 120) * r + Fraction(1, 24)
        q = q * r + Fraction(1, 6)
        q = q * r + Fraction(1, 2)
        p = r * r * q + r
        T = Fraction(float(t[kf & 127]))
        got = T * (1 + p)
        got = mp.mpf(got.numerator) / got.denominator * mp.power(2, kf >> 7)
        want = mp.exp(mp.mpf(float(x)))
        assert abs(got - want) <= 2e-16 * want, (x, float(got / want - 1))
