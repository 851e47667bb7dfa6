#!/usr/bin/env python3
"""Fit the polynomial used by the branch-free GELU core of the gfx950 forward kernel.

    Phi(-a) = 0.5*erfc(a/sqrt2) ~= exp2(-1 + a*P(a)),   a = |x| >= 0,  P of degree d

Weighted so that the ABSOLUTE error of h = Phi(-a) is minimised (that is what the result
y = x*Phi(x), Phi = h or 1-h, needs).  Prints C-style coefficients and the error of an fp32
Horner/FMA evaluation (hardware v_exp_f32 modelled as correctly rounded exp2).
"""
import sys
import numpy as np
from scipy.special import ndtr, log_ndtr

def fit(d, amax=6.0, iters=40, npts=20001):
    # Chebyshev-ish dense grid
    k = np.arange(npts)
    a = amax * 0.5 * (1 - np.cos(np.pi * k / (npts - 1)))
    a = a[1:]
    h = ndtr(-a)
    q = log_ndtr(-a) / np.log(2.0)          # log2(Phi(-a))
    target = (q + 1.0) / a                   # P(a)
    # abs error of h wrt error e in P:  h*ln2*a*e
    base_w = h * np.log(2.0) * a
    w = base_w.copy()
    V = np.vander(a, d + 1, increasing=True)
    for it in range(iters):
        c, *_ = np.linalg.lstsq(V * w[:, None], target * w, rcond=None)
        err = (V @ c - target) * base_w
        m = np.abs(err).max()
        # Lawson reweighting towards minimax
        w = w * (0.5 + 0.5*np.abs(err) / m) ** 1.0
        w = w / w.max()
    return c, m

def f32(x): return np.float32(x)

def eval_f32(c, a32):
    """fp32 Horner with FMA (emulated in float64 then rounded), returns h=exp2(fma(a,P,-1)) in f32."""
    c32 = [np.float32(v) for v in c]
    r = np.full(a32.shape, c32[-1], dtype=np.float32)
    a64 = a32.astype(np.float64)
    for v in c32[-2::-1]:
        r = (r.astype(np.float64) * a64 + np.float64(v)).astype(np.float32)
    qq = (r.astype(np.float64) * a64 - 1.0).astype(np.float32)
    return np.exp2(qq.astype(np.float64)).astype(np.float32)

if __name__ == '__main__':
    for d in (5, 6, 7, 8, 9):
        c, m = fit(d)
        a32 = np.linspace(0, 8, 2000001).astype(np.float32)
        h32 = eval_f32(c, a32)
        true = ndtr(-a32.astype(np.float64))
        abs_err = np.abs(h32.astype(np.float64) - true)
        # resulting y error in ulps of y for x>0 (Phi=1-h) and x<0 (Phi=h)
        x = a32.astype(np.float64)
        ypos = (a32 * (np.float32(1) - h32)).astype(np.float32)
        ytrue = x * (1 - true)
        ulp = np.spacing(np.abs(ytrue).astype(np.float32)).astype(np.float64)
        epos = np.abs(ypos - ytrue) / np.maximum(ulp, 1e-300)
        yneg = (-a32 * h32).astype(np.float32)
        ytn = -x * true
        ulpn = np.spacing(np.abs(ytn).astype(np.float32)).astype(np.float64)
        eneg = np.abs(yneg - ytn) / np.maximum(ulpn, 1e-300)
        print(f'd={d} fit max abs err(h)={m:.3e}  f32-eval max abs err(h)={abs_err.max():.3e} at a={a32[abs_err.argmax()]:.3f}'
              f'  y ulp err: x>0 max {epos[1:].max():.2f}  x<0 (|x|<2) max {eneg[(a32<2)&(a32>0)].max():.2f} (|x|<4) {eneg[(a32<4)&(a32>0)].max():.1f}  lead coeff {c[-1]:.3e}')
        print('   coeffs:', ', '.join(f'{np.float32(v):.9e}f' for v in c))
