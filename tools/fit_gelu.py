#!/usr/bin/env python3
"""Minimax fit of Phi(x) ~= sigmoid(x (c0 + c1 x^2 + c2 x^4)) used by the GEMM epilogues (csrc/common.h)."""
import numpy as np
from scipy import optimize, special

x = np.linspace(-9, 9, 200001)
phi_cdf = 0.5 * (1 + special.erf(x / np.sqrt(2)))
gelu = x * phi_cdf


def model(c, x):
    x2 = x * x
    with np.errstate(over="ignore"):
        return 1 / (1 + np.exp(-x * (c[0] + x2 * (c[1] + x2 * c[2]))))


r = optimize.minimize(lambda c: np.max(np.abs(x * model(c, x) - gelu)), [1.5957691216, 0.07135481627, 0.0], method="Nelder-Mead",
                      options=dict(xatol=1e-12, fatol=1e-14, maxiter=20000, maxfev=40000))
print("c =", list(r.x))
print("max |gelu err| =", np.max(np.abs(x * model(r.x, x) - gelu)), " max |Phi err| =", np.max(np.abs(model(r.x, x) - phi_cdf)))
