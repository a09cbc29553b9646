"""The decimal text of write_text (src/baproblem.rs:709-733: every f64 through Rust's `{}`) as csrc/decimal.hpp makes it
-- the functions both the host formatter and the device writer (csrc/text_kernels.hpp) run -- against an independent
implementation: CPython's repr (David Gay's shortest round-trip digits) laid out in fixed notation with Decimal."""
import math
import struct
from decimal import Decimal

import numpy as np

from city2ba_amd.baproblem import format_f64


def rust_display(x):
    x = float(x)
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    s = format(Decimal(repr(x)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    if s in ("0", "-0"):
        return "-0" if math.copysign(1.0, x) < 0 else "0"
    return s


def test_known_texts():
    """what Rust prints (the README scene's values and the edges of the format)"""
    cases = [(0.30000000000000004, "0.30000000000000004"), (1e-7, "0.0000001"), (-0.0, "-0"), (0.0, "0"), (1.0, "1"),
             (-2.5e-10, "-0.00000000025"), (123456789.125, "123456789.125"), (1e22, "10000000000000000000000"),
             (1e23, "100000000000000000000000"),             # to_chars(fixed) prints 99999999999999991611392 here
             (1.2345678901234567e30, "1234567890123456700000000000000"), (9007199254740993.0, "9007199254740992"),
             (5e-324, "0." + "0" * 323 + "5"), (float("inf"), "inf"), (float("-inf"), "-inf"), (float("nan"), "NaN"),
             (1.7976931348623157e308, "17976931348623157" + "0" * 292), (2.2250738585072014e-308, "0." + "0" * 307 + "22250738585072014"),
             (0.1, "0.1"), (100.0, "100"), (1234.5, "1234.5"), (-1e-5, "-0.00001"), (4.35, "4.35"), (0.000001, "0.000001")]
    assert format_f64([v for v, _ in cases]) == [s for _, s in cases]


def test_against_python_repr_over_every_magnitude():
    rng = np.random.default_rng(11)
    vals = [
        rng.integers(0, 2**64, 120_000, dtype=np.uint64).view(np.float64),             # any bit pattern: every exponent, subnormals, NaNs
        rng.uniform(-1, 1, 60_000), rng.uniform(-1000, 1000, 60_000),
        rng.integers(0, 2_000_000, 40_000) / 1000.0,                                       # short decimals
        rng.integers(0, 10**11, 40_000).astype(np.float64),                                # integers
        np.ldexp(rng.integers(0, 4096, 40_000).astype(np.float64), -rng.integers(0, 60, 40_000)),   # dyadic: exact ties
        rng.uniform(-1, 1, 60_000) * 10.0 ** rng.integers(-30, 40, 60_000),
        np.ldexp(1.0, np.arange(-1074, 1024)), np.ldexp(1.0, np.arange(-1074, 1024)) * (1 + 2.0**-52),      # every binade's edge
        np.ldexp(1.0 - 2.0**-53, np.arange(-1021, 1024)), 10.0 ** np.arange(-323, 309),
    ]
    v = np.concatenate([np.asarray(x, dtype=np.float64) for x in vals])
    got = format_f64(v)
    assert len(got) == len(v)
    bad = [(x, g, rust_display(x)) for x, g in zip(v.tolist(), got) if g != rust_display(x)]
    assert not bad, bad[:5]


def test_texts_read_back_to_the_same_bits():
    rng = np.random.default_rng(12)
    v = rng.integers(0, 2**64, 50_000, dtype=np.uint64).view(np.float64)
    v = v[np.isfinite(v)]
    back = np.array([float(s) for s in format_f64(v)])
    assert np.array_equal(back.view(np.uint64), v.view(np.uint64))
    # and nothing shorter does: dropping the last digit must change the value (checked on values with a fraction)
    w = rng.uniform(-1000, 1000, 5_000)
    for x, s in zip(w.tolist(), format_f64(w)):
        if "." in s and len(s.rstrip("0")) > 3:
            assert float(s[:-1]) != x, (x, s)
    assert struct.pack("<d", float(format_f64([-0.0])[0])) == struct.pack("<d", -0.0)


# ---- the other direction: from_file_text's numbers (src/baproblem.rs:580-629: nom `double` = str::parse::<f64>) -----
def test_parser_against_python_float_on_every_spelling():
    """c2b_parse_f64 (csrc/decimal.hpp: Clinger's exact case, then Eisel-Lemire) against CPython's float() -- David Gay's
    correctly rounded strtod, an independent implementation: shortest and 17-digit forms of random bit patterns,
    exponent spellings, integers up to 19 digits, and decimal strings lying EXACTLY half way between two doubles."""
    from city2ba_amd.baproblem import parse_f64
    rng = np.random.default_rng(13)
    toks = []
    bits = rng.integers(0, 2**64, 60_000, dtype=np.uint64).view(np.float64)
    bits = bits[np.isfinite(bits)]
    toks += [repr(float(x)) for x in bits[:20_000]]                                   # shortest, exponent form for large / small
    toks += ["%.17g" % x for x in bits[20_000:40_000]] + ["%+.16E" % x for x in bits[40_000:]]
    toks += format_f64(rng.uniform(-1000, 1000, 20_000)) + format_f64(10.0 ** rng.integers(-320, 308, 3_000) * rng.uniform(1, 10, 3_000))
    toks += [str(int(x) >> int(s)) for x, s in zip(rng.integers(0, 2**63, 20_000), rng.integers(0, 63, 20_000))]
    toks += ["%de%d" % (int(x) >> int(s), int(e)) for x, s, e in zip(rng.integers(0, 2**63, 20_000), rng.integers(0, 63, 20_000), rng.integers(-360, 340, 20_000))]
    toks += ["%d.%de%+d" % (a, b, e) for a, b, e in zip(rng.integers(0, 10**5, 10_000), rng.integers(0, 10**11, 10_000), rng.integers(-30, 30, 10_000))]
    for m, e in zip(rng.integers(2**52, 2**53, 10_000), rng.integers(-3, 9, 10_000)):   # (2m + 1) * 2^e written out exactly: a tie
        m, e = int(m), int(e)
        toks.append(str((2 * m + 1) << e) if e >= 0 else format(Decimal(2 * m + 1) / Decimal(2 ** -e), "f"))
    toks += ["0", "-0", "0.0", "-0.000e5", "1e400", "-1e400", "1e-400", "4.9e-324", "2.4703282292062327e-324", "2.4703282292062328e-324",
             "1.7976931348623157e308", "1.7976931348623158e308", "1.7976931348623159e308", "9007199254740993", "9007199254740992.5",
             "0.000000000000000000000000000001", "100000000000000000000000", "+1.5", ".5", "5.", "1E5"]
    vals, st = parse_f64(toks)
    assert (st == 1).sum() == 0
    ok = st == 0
    assert ok.sum() > 0.9 * len(toks)                          # `unsure` only for the > 19-digit ties
    want = np.array([float(t) for t in toks])
    bad = [(t, v, w) for t, v, w, o in zip(toks, vals.tolist(), want.tolist(), ok.tolist()) if o and np.float64(v).view(np.uint64) != np.float64(w).view(np.uint64)]
    assert not bad, bad[:5]
    assert all(len(t.replace(".", "").replace("-", "").lstrip("0")) > 19 for t, s in zip(toks, st.tolist()) if s == 2)
    # spellings the parser leaves to the host's strtod
    vals, st = parse_f64(["NaN", "inf", "0x1p3", "1e", "1.5.2", "1-2", "--1", "e5", ".", "1_000"])
    assert (st == 1).all()


def test_format_then_parse_is_the_identity():
    from city2ba_amd.baproblem import parse_f64
    rng = np.random.default_rng(14)
    v = rng.integers(0, 2**64, 200_000, dtype=np.uint64).view(np.float64)
    v = v[np.isfinite(v)]
    back, st = parse_f64(format_f64(v))
    assert (st == 0).all() and np.array_equal(back.view(np.uint64), v.view(np.uint64))
