"""Fit and check the coefficients of erf_fast and (with --gelu) gelu16_wt (csrc/encoder.hip): Chebyshev fits of erf(x)/x in x^2 on [0, 0.921875]
and of log2(erfc(t)) on [0.921875, 4], evaluated in emulated float32 fma arithmetic against scipy over 620k points."""
import sys

import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C, polynomial as Pn
from mpmath import mp, erfc, log
mp.dps = 40
def nodes(a, b, n):
    k = np.arange(n); return 0.5*(a+b) + 0.5*(b-a)*np.cos((2*k+1)*np.pi/(2*n))
T = 0.921875
s = nodes(0, T*T, 400); x = np.sqrt(s)
cA = C.Chebyshev.fit(s, erf(x)/x, 6, domain=[0, T*T]).convert(kind=Pn.Polynomial).coef
t = nodes(T, 4.0, 600)
y = np.array([float(log(erfc(mp.mpf(v)))/log(2)) for v in t])
cB = C.Chebyshev.fit(t, y, 8, domain=[T, 4.0]).convert(kind=Pn.Polynomial).coef
f32 = np.float32
def fma(a, b, c): return f32(np.float64(a)*np.float64(b) + np.float64(c))
fma = np.vectorize(fma, otypes=[np.float32])
def erf32(x):
    x = x.astype(f32); t = np.minimum(np.abs(x), f32(4.0)).astype(f32); s = (x*x).astype(f32)
    ra = np.full_like(x, f32(cA[-1]))
    for k in cA[-2::-1]: ra = fma(ra, s, f32(k))
    ra = (x * ra).astype(f32)
    rb = np.full_like(x, f32(cB[-1]))
    for k in cB[-2::-1]: rb = fma(rb, t, f32(k))
    e = np.exp2(rb.astype(np.float64)).astype(f32)
    rb = np.copysign((f32(1.0) - e).astype(f32), x)
    return np.where(t <= f32(T), ra, rb)
xs = np.concatenate([np.linspace(-6, 6, 600001), np.logspace(-8, 0, 20001), [0.0, T, np.nextafter(f32(T), f32(1))]]).astype(f32)
got = erf32(xs).astype(np.float64); want = erf(xs.astype(np.float64))
err = np.abs(got - want)
print("max abs err", err.max(), "at", xs[err.argmax()])
nz = np.abs(want) > 0
print("max rel err", (err[nz]/np.abs(want[nz])).max(), "ulps", (err[nz]/np.abs(want[nz])).max()/2**-24)
print("A:", ", ".join("%.9ef" % v for v in cA))
print("B:", ", ".join("%.9ef" % v for v in cB))


def fit_gelu():
    """gelu16_wt: gelu(x) = max(x,0) - 0.5|x| * 2^Q(min(|x|, 5.75)), Q = Chebyshev fit of log2(erfc(t/sqrt 2)) on [0, 5.75]."""
    from mpmath import sqrt
    CAP = 5.75
    t = nodes(0, CAP, 800)
    y = np.array([float(log(erfc(mp.mpf(v)/sqrt(2)))/log(2)) for v in t])
    xs = np.concatenate([np.linspace(-8, 8, 400001), np.logspace(-8, 0, 20001), -np.logspace(-8, 0, 20001)]).astype(f32)
    want = 0.5*xs.astype(np.float64)*(1+erf(xs.astype(np.float64)/np.sqrt(2)))
    for deg in (9, 10, 11):
        cQ = C.Chebyshev.fit(t, y, deg, domain=[0, CAP]).convert(kind=Pn.Polynomial).coef
        tt = np.minimum(np.abs(xs), f32(CAP)).astype(f32)
        r = np.full_like(xs, f32(cQ[-1]))
        for k in cQ[-2::-1]: r = fma(r, tt, f32(k))
        e = np.exp2(r.astype(np.float64)).astype(f32)
        g = fma((f32(-0.5)*np.abs(xs)).astype(f32), e, np.maximum(xs, f32(0)))
        err = np.abs(g.astype(np.float64) - want)
        big = np.abs(want) > 1e-3
        print(deg, "max abs err %.3e at x=%.4f" % (err.max(), xs[err.argmax()]), " max rel err where |gelu|>1e-3: %.3e" % (err[big]/np.abs(want[big])).max())
        print("  Q:", ", ".join("%.9ef" % v for v in cQ))


if "--gelu" in sys.argv:
    fit_gelu()
