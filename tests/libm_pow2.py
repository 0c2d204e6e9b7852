"""A Python transliteration of `exp2_libm` (s-rack_amd/csrc/modules.hip.h): the x86-64 FMA build of glibc's pow for the base 2.0,
operation for operation, with exact fused multiply-adds (rational arithmetic, one rounding).  Test infrastructure: tests/test_oracle.py
runs it against the host libm's pow — the function the oracle (and an unoptimised build of the reference) calls for
`2.0_f64.powf(e)`, oscillator.rs:45 — and checks that the device header holds the same constants and table."""
import math
import os
import re
import struct
from fractions import Fraction

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "s-rack_amd", "csrc", "modules.hip.h")


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # Fraction -> float rounds to nearest even: one rounding


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _dbl(u):
    return struct.unpack("<d", struct.pack("<Q", u & 0xFFFFFFFFFFFFFFFF))[0]


def header_table():
    """the 256 words of kLibmExpTab as the device header spells them"""
    text = open(HEADER).read()
    body = text[text.index("kLibmExpTab[256] = {"):]
    body = body[:body.index("};")]
    return [int(w, 16) for w in re.findall(r"0x[0-9a-f]{16}", body)]


LHI, LLO = float.fromhex("0x1.62e42fefa39efp-1"), float.fromhex("0x1.abc9e3b398000p-56")
INVLN2N, SHIFT = float.fromhex("0x1.71547652b82fep+7"), float.fromhex("0x1.8p52")
NEGLN2HI, NEGLN2LO = float.fromhex("-0x1.62e42fefa0000p-8"), float.fromhex("-0x1.cf79abc9e3b3ap-47")
C2, C3, C4, C5 = (float.fromhex(h) for h in ("0x1.ffffffffffdbdp-2", "0x1.555555555543cp-3", "0x1.55555cf172b91p-5", "0x1.1111167a4d017p-7"))


def exp2_libm(e, tab):
    ehi = e * LHI
    elo = fma(e, LLO, fma(LHI, e, -ehi))
    abstop = (_bits(ehi) >> 52) & 0x7FF
    if not (0 <= abstop - 0x3C9 <= 0x3E):
        topy = (_bits(e) >> 52) & 0x7FF
        if topy < 0x3BE:
            return 1.0 + e
        if topy < 0x43E and abstop < 0x3C9:
            return 1.0 + ehi
        return math.pow(2.0, e)   # (the device hands these to ocml's pow: inf, 0, NaN — and |e ln 2| >= 512, which is no pitch)
    kds = fma(ehi, INVLN2N, SHIFT)
    ki = _bits(kds)
    kd = kds - SHIFT
    r = fma(kd, NEGLN2LO, fma(kd, NEGLN2HI, ehi))
    r = elo + r
    idx = 2 * (ki & 127)
    tail = _dbl(tab[idx])
    sbits = (tab[idx + 1] + (ki << 45)) & 0xFFFFFFFFFFFFFFFF
    r2 = r * r
    a = fma(r, C3, C2)
    b = r + tail
    c = fma(r, C5, C4)
    tmp = fma(a, r2, b)
    tmp = fma(c, r2 * r2, tmp)
    scale = _dbl(sbits)
    return fma(tmp, scale, scale)
