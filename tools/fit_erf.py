#!/usr/bin/env python3
"""Coefficients of csrc/itr_common.h::gelu_erf: erf(t) = 1 - exp(-t Q(t)) on [0, 4], Q = degree-7 polynomial fitted to
-ln(erfc t) / t by iteratively re-weighted least squares on the erf error (a Remez-like equalisation), then rounded to fp32 and
checked in emulated fp32 arithmetic (Horner with fp32 rounding after every step, exp2 in fp32) against scipy's float64 erf.

    python tools/fit_erf.py            # prints the coefficients (lowest degree first) and the error figures
"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf, erfc

T, DEG = 4.0, 7
t = np.linspace(1e-6, T, 400001)
g = -np.log(erfc(t)) / t
w = t * np.exp(-t * g) + 1e-3
x = 2 * t / T - 1
coef = C.chebfit(x, g, DEG, w=w)
for _ in range(60):
    err = (1 - np.exp(-t * C.chebval(x, coef))) - erf(t)
    w = w * (1 + 4 * np.abs(err) / np.abs(err).max())
    coef = C.chebfit(x, g, DEG, w=w)
P = np.polynomial.Polynomial(C.cheb2poly(coef))(np.polynomial.Polynomial([-1, 2 / T]))
c = [float(np.float32(v)) for v in P.coef]
print("Q coefficients (t^0 .. t^%d):" % DEG, ", ".join("%.9ef" % v for v in c))


def gelu32(xs):
    tf = np.minimum(np.abs(xs) * np.float32(0.70710678118654752440), np.float32(T)).astype(np.float32)
    acc = np.float32(c[-1]) * np.ones_like(tf)
    for k in range(len(c) - 2, -1, -1):
        acc = np.float32(acc * tf + np.float32(c[k]))
    r = (np.float32(1) - np.exp2(np.float32(np.float32(acc * tf) * np.float32(-1.4426950408889634))).astype(np.float32)).astype(np.float32)
    hx = np.float32(0.5) * xs
    return np.float32(hx * np.copysign(r, xs) + hx), r


xs = np.float32(np.linspace(-8, 8, 1600001))
gel, r = gelu32(xs)
ref = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
tt = np.minimum(np.abs(xs.astype(np.float64)) / np.sqrt(2), 1e9)
print("max |erf error| (fp32 arithmetic)  %.3g" % np.abs(r.astype(np.float64) - erf(tt)).max())
print("max |gelu error| over [-8, 8]      %.3g" % np.abs(gel - ref).max())
