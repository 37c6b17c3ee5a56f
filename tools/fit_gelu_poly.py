#!/usr/bin/env python
"""Fit of the two odd polynomials behind the erf-GELU of clipa_amd/csrc/common.h (gelu_cdf2, gelu_grad2).

    Phi(x)  - 1/2 = erf(x / sqrt 2) / 2              ~ x * P(x^2),  deg P = 9     (forward:  gelu(x)  = x Phi(x))
    gelu'(x) - 1/2 = erf(x / sqrt 2) / 2 + x phi(x)   ~ x * Q(x^2),  deg Q = 10    (backward: d gelu / dx)

on |x| <= X0 = 4.5 (the argument is clamped there; Phi is clamped to [0, 1]).  Weighted least squares on Chebyshev nodes
of t = x^2 with a few Lawson re-weighting rounds (weights: the error of x Phi(x) resp. of gelu'), converted to monomials
in t for Horner evaluation in fp32.  `python tools/fit_gelu_poly.py` prints the coefficients (highest degree first, the
order of the Horner chain in common.h) and the error of the fp32 evaluation against scipy's erf.
tests/test_host_cpu.py::test_gelu_polynomials re-evaluates the coefficients it parses from common.h.
"""
import numpy as np
import scipy.special as sp
from numpy.polynomial import chebyshev as C, polynomial as Pn

X0 = 4.5


def fit_odd(fun_over_x, deg, wfun, iters=30):
    u = np.cos(np.pi * (np.arange(8000) + 0.5) / 8000)
    t = (u + 1) / 2 * X0 * X0
    xs = np.sqrt(t) + 1e-9
    y = fun_over_x(xs)
    w0 = wfun(xs)
    w = w0.copy()
    for _ in range(iters):
        c = C.chebfit(u, y, deg, w=w)
        e = np.abs((C.chebval(u, c) - y) * w0)
        w = w * (1 + 4 * e / e.max())
        w = w / w.max() * w0.max()
    p_u = C.cheb2poly(c)
    pt, lin = np.zeros(1), np.array([-1.0, 2 / (X0 * X0)])
    for k, ck in enumerate(p_u):
        term = np.array([1.0])
        for _ in range(k):
            term = Pn.polymul(term, lin)
        pt = Pn.polyadd(pt, ck * term)
    return pt          # ascending powers of t


def horner32(coef_desc, t):
    r = np.full_like(t, np.float32(coef_desc[0]))
    for ck in coef_desc[1:]:
        r = np.float32(r * t + np.float32(ck))
    return r


def gelu_cdf32(x, coef_desc):
    xc = np.clip(x, -X0, X0).astype(np.float32)
    p = horner32(coef_desc, np.float32(xc * xc))
    return np.clip(np.float32(np.float32(xc * p) + np.float32(0.5)), 0, 1).astype(np.float32)


def gelu_grad32(x, coef_desc):
    xc = np.clip(x, -X0, X0).astype(np.float32)
    p = horner32(coef_desc, np.float32(xc * xc))
    return np.float32(np.float32(xc * p) + np.float32(0.5))


def Phi(x):
    return 0.5 * (1 + sp.erf(x / np.sqrt(2)))


def phi(x):
    return np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


if __name__ == "__main__":
    xs = np.linspace(-12, 12, 1200001).astype(np.float32)
    cf = fit_odd(lambda x: 0.5 * sp.erf(x / np.sqrt(2)) / x, 9, lambda x: x ** 2)[::-1]
    cb = fit_odd(lambda x: (0.5 * sp.erf(x / np.sqrt(2)) + x * phi(x)) / x, 10, lambda x: x)[::-1]
    print("forward  P (highest degree first):", ", ".join("%.9ef" % c for c in cf))
    print("backward Q (highest degree first):", ", ".join("%.9ef" % c for c in cb))
    x64 = xs.astype(np.float64)
    ef = np.abs(np.float32(xs * gelu_cdf32(xs, cf.astype(np.float32))) - x64 * Phi(x64))
    eb = np.abs(gelu_grad32(xs, cb.astype(np.float32)) - (Phi(x64) + x64 * phi(x64)))
    for name, e in (("x Phi(x)", ef), ("gelu'(x)", eb)):
        print(f"{name}: max |err| {e.max():.2e} at x = {xs[e.argmax()]:.2f}; |x| < 3: {e[np.abs(xs) < 3].max():.2e}; "
              f"|x| <= 4.5: {e[np.abs(xs) <= 4.5].max():.2e}")
