#!/usr/bin/env python3
"""Coefficients of the one-transcendental erf of the f16 Decision-Transformer kernels (busca_amd/csrc/dt_kernel.hip.inc, Prec<1>::gelu):
erf(s) = sign(s) (1 - 2^(-|s| P(|s|))) with P a polynomial fitted to -log2(erfc(s)) / s on [0, 4] (reweighted least squares towards the
minimax of the ERF error), and the error of the float32 evaluation over [0, 6].  python tools/fit_erf.py [degree]"""
import sys

import numpy as np
from scipy.special import erf, erfc

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
X = 4.0
xs = np.linspace(1e-4, X, 200001)
q = -np.log2(erfc(xs)) / xs
w = np.ones_like(xs)
for _ in range(30):
    c = np.polynomial.chebyshev.chebfit(2 * xs / X - 1, q, deg, w=w)
    err = np.abs((1 - 2.0 ** (-xs * np.polynomial.chebyshev.chebval(2 * xs / X - 1, c))) - erf(xs))
    w = w * (1 + 2 * err / err.max())
co = np.polynomial.Polynomial(np.polynomial.chebyshev.cheb2poly(c))(np.polynomial.Polynomial([-1, 2 / X])).coef.astype(np.float32)
x32 = np.linspace(0, 6, 600001).astype(np.float32)
ax = np.minimum(x32, np.float32(X))
acc = np.full_like(ax, co[-1])
for k in co[-2::-1]:
    acc = (acc * ax + k).astype(np.float32)
r = (np.float32(1) - np.exp2((-ax * acc).astype(np.float32)).astype(np.float32)).astype(np.float32)
print("degree %d, coefficients (constant first): %s" % (deg, [float(v) for v in co]))
print("max |erf error| of the float32 evaluation on [0, 6]: %.3g" % np.abs(r - erf(x32.astype(np.float64))).max())
g = 0.5 * x32.astype(np.float64) * np.sqrt(2) * (1 + r)          # gelu at v = s sqrt(2)
ge = 0.5 * x32.astype(np.float64) * np.sqrt(2) * (1 + erf(x32.astype(np.float64)))
print("max |gelu error| for v >= 0: %.3g; for v <= 0: %.3g" % (np.abs(g - ge).max(), np.abs(0.5 * x32 * np.sqrt(2) * ((1 - r) - erfc(x32.astype(np.float64)))).max()))
