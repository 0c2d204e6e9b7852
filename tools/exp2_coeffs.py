"""Coefficients of modules.hip.h's polynomial 2^f on |f| <= 1/2 (mpmath, 60 digits): the Chebyshev interpolant of the given degree in
monomial form, rounded to f64, and its error as evaluated in f64 by the kernel's own scheme.  usage: python tools/exp2_coeffs.py [degree]"""
import sys
import mpmath as mp
import numpy as np
mp.mp.dps = 60
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = deg + 1
nodes = [mp.cos(mp.pi * (j + mp.mpf(1) / 2) / n) for j in range(n)]
vals = [mp.power(2, x / 2) for x in nodes]
a = [(mp.mpf(2) / n) * sum(vals[j] * mp.cos(k * mp.pi * (j + mp.mpf(1) / 2) / n) for j in range(n)) for k in range(n)]
a[0] /= 2
# Chebyshev -> monomial in x
T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
for k in range(2, n):
    t = [mp.mpf(0)] + [2 * c for c in T[k - 1]]
    for i, c in enumerate(T[k - 2]): t[i] -= c
    T.append(t)
mono = [mp.mpf(0)] * n
for k in range(n):
    for i, c in enumerate(T[k]): mono[i] += a[k] * c
coef = [mono[k] * mp.mpf(2) ** k for k in range(n)]  # in f = x / 2
c64 = [float(c) for c in coef]
for k, c in enumerate(c64): print(f"c{k} = {c!r}  ({c.hex()})")
# error of the f64-rounded coefficients, exact evaluation
worst = mp.mpf(0)
for i in range(-2000, 2001):
    f = mp.mpf(i) / 4000
    p = sum(mp.mpf(c) * f ** k for k, c in enumerate(c64))
    worst = max(worst, abs(p / mp.power(2, f) - 1))
print("max relative error of the rounded coefficients (exact arithmetic):", mp.nstr(worst, 5))
