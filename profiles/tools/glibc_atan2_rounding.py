"""Is this image's glibc (2.35) atan2 / acos CORRECTLY ROUNDED? If it were, any correctly rounded device routine (a double-double evaluation with a
rounding test) would match it bit for bit and restating glibc's own algorithm - as done for pow, pt_pow.h - would be unnecessary.
Measured here (50,000 points on the unit sphere like sphere.rs:57-60's arguments, mpmath at 300 bits as the judge, the result compared with both
neighbouring doubles): atan2 is NOT the nearest double in 48 cases, acos in 35 - about 0.1 %. So it is not, and only an operation-for-operation
restatement of the variant libm dispatches to on the reference's machine can reproduce it. usage: python3 profiles/tools/glibc_atan2_rounding.py [n]"""
import ctypes, numpy as np, mpmath as mp, sys, math
libm = ctypes.CDLL("libm.so.6")
libm.atan2.restype = ctypes.c_double; libm.atan2.argtypes = [ctypes.c_double, ctypes.c_double]
libm.acos.restype = ctypes.c_double; libm.acos.argtypes = [ctypes.c_double]
mp.mp.prec = 300
def is_nearest(g, exact):
    e = abs(mp.mpf(g) - exact)
    return e <= abs(mp.mpf(math.nextafter(g, math.inf)) - exact) and e <= abs(mp.mpf(math.nextafter(g, -math.inf)) - exact)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
bad_a = bad_c = 0
for i in range(n):
    y, x, c = -d[i, 2], d[i, 0], d[i, 1]
    if not is_nearest(libm.atan2(y, x), mp.atan2(mp.mpf(y), mp.mpf(x))): bad_a += 1
    if not is_nearest(libm.acos(c), mp.acos(mp.mpf(c))): bad_c += 1
print("glibc atan2 not the nearest double:", bad_a, "of", n, "; acos:", bad_c)
