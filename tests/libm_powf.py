"""A Python transliteration of `powf_libm` (s-rack_amd/csrc/modules.hip.h): the x86-64 FMA build of glibc 2.35's powf (sysdeps/ieee754/flt-32/e_powf.c,
Szabolcs Nagy's algorithm) for a positive finite x and a finite non-zero y, operation for operation as the host executes it (disassembled: which
products are contracted into fused multiply-adds is the compiler's choice, and part of the result), with exact fused multiply-adds (rational arithmetic,
one rounding).  Test infrastructure: tests/test_oracle.py runs it against the host libm's powf — the function the oracle (and the reference:
`f32::powf`, math.rs:203-205, sample.rs) calls — and checks that the device header holds the same constants and tables."""
import os
import re
import struct
from fractions import Fraction

import numpy as np

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "s-rack_amd", "csrc", "modules.hip.h")


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # Fraction -> float rounds to nearest even: one rounding


def _f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _bits_f32(u):
    return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]


def _dbl(u):
    return struct.unpack("<d", struct.pack("<Q", u & 0xFFFFFFFFFFFFFFFF))[0]


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def header_tables():
    """(kPowfLog2Tab as 16 pairs of words, kExp2fTab as 32 words) as the device header spells them"""
    text = open(HEADER).read()
    def words(name, n):
        body = text[text.index(name):]
        body = body[:body.index("};")]
        w = [int(x, 16) for x in re.findall(r"0x[0-9a-f]{16}", body)]
        assert len(w) == n, (name, len(w))
        return w
    lg = words("kPowfLog2Tab[16][2] = {", 32)
    return [(lg[2 * i], lg[2 * i + 1]) for i in range(16)], words("kExp2fTab[32] = {", 32)


A = [float.fromhex(h) for h in ("0x1.27616c9496e0bp-2", "-0x1.71969a075c67ap-2", "0x1.ec70a6ca7baddp-2", "-0x1.7154748bef6c8p-1", "0x1.71547652ab82bp+0")]
C = [float.fromhex(h) for h in ("0x1.c6af84b912394p-5", "0x1.ebfce50fac4f3p-3", "0x1.62e42ff0c52d6p-1")]
SHIFT = float.fromhex("0x1.8p+52") / 32.0


def powf_libm(x, y, log2_tab, exp2f_tab):
    """x: a positive finite float32 (subnormals included), y: a finite non-zero float32 -> float32 (as a Python float)"""
    ix = _f32_bits(x)
    if ix < 0x00800000:   # subnormal x: normalise
        ix = _f32_bits(float(np.float32(x) * np.float32(2.0 ** 23))) & 0x7FFFFFFF
        ix = (ix - (23 << 23)) & 0xFFFFFFFF
    # log2_inline
    tmp = (ix - 0x3F330000) & 0xFFFFFFFF
    i = (tmp >> 19) % 16
    top = tmp & 0xFF800000
    iz = (ix - top) & 0xFFFFFFFF
    k = top if top < 0x80000000 else top - (1 << 32)
    k >>= 23                                  # arithmetic shift
    invc, logc = _dbl(log2_tab[i][0]), _dbl(log2_tab[i][1])
    z = float(_bits_f32(iz))
    r = fma(z, invc, -1.0)
    y0 = logc + float(k)
    yy = fma(r, A[0], A[1])
    p = fma(r, A[2], A[3])
    r2 = r * r
    q = fma(r, A[4], y0)
    r4 = r2 * r2
    q = fma(r2, p, q)
    logx = fma(yy, r4, q)
    ylogx = float(np.float32(y)) * logx
    if (_bits(ylogx) >> 47 & 0xFFFF) >= (_bits(126.0) >> 47):
        if ylogx > float.fromhex("0x1.fffffffd1d571p+6"):
            return float("inf")
        if ylogx <= -150.0:
            return 0.0
    # exp2_inline (sign_bias 0)
    kd = ylogx + SHIFT
    ki = _bits(kd)
    kd -= SHIFT
    r = ylogx - kd
    t = (exp2f_tab[ki % 32] + (ki << 47)) & 0xFFFFFFFFFFFFFFFF
    s = _dbl(t)
    zz = fma(r, C[0], C[1])
    r2 = r * r
    yv = fma(r, C[2], 1.0)
    yv = fma(zz, r2, yv)
    return float(np.float32(yv * s))
