"""Generates the fp64 polynomial used by erfcx_fast_d (blues_amd/csrc/device_common.h):
erfc(x) = exp(-x^2) * P(u),  t = 1/(1 + 0.3 x),  u = A t + B in [-1, 1],  x in [0, 6]  (Chebyshev fit, monomial form).
Max relative error ~5e-15 over the interval (checked below against scipy.special.erfcx)."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfcx

p, xmax, deg = 0.3, 6.0, 16
tmin = 1 / (1 + p * xmax)
k = np.arange(600); u = np.cos(np.pi * (k + 0.5) / 600)
t = 0.5 * (1 + tmin) + 0.5 * (1 - tmin) * u
c = C.chebfit(u, erfcx((1 / t - 1) / p), deg)
mono = C.cheb2poly(c)
xt = np.linspace(0, xmax, 400001)
ut = (1 / (1 + p * xt) - 0.5 * (1 + tmin)) / (0.5 * (1 - tmin))
acc = np.zeros_like(ut) + mono[-1]
for a in mono[-2::-1]:
    acc = acc * ut + a
print("// max rel err %.2e on [0,%g]" % (np.abs(acc / erfcx(xt) - 1).max(), xmax))
print("#define ERFCX_A %.17g\n#define ERFCX_B %.17g" % (2 / (1 - tmin), -(1 + tmin) / (1 - tmin)))
print("static __device__ const double ERFCX_C[%d] = {\n    %s};" % (deg + 1, ",\n    ".join("%.17e" % a for a in mono)))
