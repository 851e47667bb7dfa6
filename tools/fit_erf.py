#!/usr/bin/env python3
"""Fit the two branch-free pieces of the fp32 erf used by the precise (fp32 I/O) GELU:
   |z| <  T : erf(z) = z + z*R(z^2)                      R of degree 5
   |z| >= T : erf(z) = sign(z) * (1 - exp(-(t + t*Q(t))))    t = |z|, Q of degree 6
and report the error of an fp32 Horner/FMA evaluation in ulps of erf (hardware exp modelled as exact exp2 of the
fp32 argument with the product error compensated, as the kernel does)."""
import numpy as np
from scipy.special import erf, erfc

T = 0.921875

def lawson(V, target, wbase, iters=60):
    w = wbase.copy()
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * w[:, None], target * w, rcond=None)
        err = np.abs((V @ c - target) * wbase)
        w = w * (0.5 + 0.5 * err / err.max())
        w /= w.max()
    return c, err.max()

def cheb(a, b, n):
    k = np.arange(n)
    return a + (b - a) * 0.5 * (1 - np.cos(np.pi * (k + 0.5) / n))

# small branch
z = cheb(1e-4, T, 4001); s = z * z
Rt = erf(z) / z - 1.0
cS, eS = lawson(np.vander(s, 6, increasing=True), Rt, z / erf(z))       # relative error of erf
# large branch
t = cheb(T, 4.2, 6001)
Qt = -np.log(erfc(t)) / t - 1.0
# error of erf = erfc * t * dQ (abs) ; relative to erf ~ 1
cL, eL = lawson(np.vander(t, 7, increasing=True), Qt, erfc(t) * t / erf(t))
print('small: fit rel err %.3e' % eS, '\n  ', ', '.join('%.9ef' % np.float32(v) for v in cS))
print('large: fit rel err %.3e' % eL, '\n  ', ', '.join('%.9ef' % np.float32(v) for v in cL))

def f32(x): return np.asarray(x, dtype=np.float32)
def fma(a, b, c): return f32(a.astype(np.float64) * b.astype(np.float64) + np.asarray(c, dtype=np.float64))

def erf_f32(zz):
    zz = f32(zz); a = np.abs(zz); s = f32(a.astype(np.float64) ** 2)
    cs = [np.float32(v) for v in cS]; cl = [np.float32(v) for v in cL]
    r = np.full(a.shape, cs[5], np.float32)
    for v in cs[4::-1]: r = fma(r, s, v)
    small = fma(r, a, a)
    q = np.full(a.shape, cl[6], np.float32)
    for v in cl[5::-1]: q = fma(q, a, v)
    p = fma(q, a, a)                                   # p = t + t*Q(t)
    L = np.float32(1.4426950408889634); Llo = np.float32(1.4426950408889634 - float(np.float32(1.4426950408889634)))
    u = f32(p.astype(np.float64) * L)                  # rounded product
    e = fma(p, np.full(a.shape, L), -u.astype(np.float64))      # its rounding error
    e = fma(p, np.full(a.shape, Llo), e)
    ex = f32(np.exp2(-u.astype(np.float64)))           # v_exp_f32(-u), modelled exact
    ex = fma(-ex, f32(e * np.float32(0.6931471805599453)), ex)   # * (1 - e*ln2)
    large = f32(1.0 - ex.astype(np.float64))
    out = np.where(a < np.float32(T), small, large)
    return np.copysign(out, zz)

zz = np.concatenate([np.linspace(0, 4.2, 4000001), np.random.default_rng(0).normal(size=2000000) * 1.2]).astype(np.float32)
got = erf_f32(zz).astype(np.float64); ref = erf(zz.astype(np.float64))
ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
err = np.abs(got - ref) / np.maximum(ulp, 1e-300)
m = np.abs(zz) > 1e-30
print('max ulp err small branch %.3f, large branch %.3f' % (err[m & (np.abs(zz) < T)].max(), err[m & (np.abs(zz) >= T)].max()))
