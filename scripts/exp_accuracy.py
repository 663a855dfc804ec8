"""ulp error of the K build's exp against 40-digit mpmath, through gpmi_rbf (1-D inputs whose
squared distances sweep the argument range; the argument coef*s is formed in fp64 exactly as
the kernel forms it)."""
import os, sys
import numpy as np
from mpmath import mp, mpf, exp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
mp.dps = 40
ctx = GPContext(0)
rng = np.random.default_rng(5)
for hi, ell in ((6.0, 1.0), (37.0, 1.0), (1.2, 1.0)):
    a = rng.uniform(0, hi, (384, 1)); b = rng.uniform(0, hi, (384, 1))
    K = ctx.rbf(a, b, 1.0, ell)
    s = (a - b.T) ** 2                     # d = 1: one term, no summation
    arg = (-.5 * (1 / (ell * ell))) * s
    worst = 0.0; tot = 0.0; cnt = 0
    for i in range(0, 384, 3):
        for j in range(384):
            x = float(arg[i, j])
            if x < -700: continue
            t = exp(mpf(x))
            k = K[i, j]
            u = np.spacing(k)
            e = abs(float((mpf(k) - t) / mpf(u)))
            worst = max(worst, e); tot += e; cnt += 1
    print("args in [%.1f, 0]: %d samples, max error %.3f ulp, mean %.3f ulp" % (arg.min(), cnt, worst, tot / cnt))
