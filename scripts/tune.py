"""A/B tuning knobs on the GPU: python scripts/tune.py N [n_test]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O  # noqa: E402
from gaussian_process_amd import GPContext  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
configs = sys.argv[3:] or ["la=0,nb=512", "la=1,nb=512", "la=1,nb=256", "la=1,nb=1024", "la=1,nb=2048"]
X, y, Xs = O.synthetic_problem(N, 8, n)
ctx = GPContext(0)
ctx.set_train(X, y)
ctx.set_test(Xs)
ref = None
for cfg in configs:
    kv = dict(p.split("=") for p in cfg.split(","))
    ctx.set_option("lookahead", int(kv.get("la", 1)))
    ctx.set_option("nb", int(kv.get("nb", 512)))
    ctx.set_option("ramp", int(kv.get("ramp", 0)))
    if "pad" in kv:
        ctx.set_option("ld_pad", int(kv["pad"]))
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        lml = ctx.factorize(1.0, 2.0, 5e-4)
        t1 = time.perf_counter()
        tm = ctx.timers()
        mu, var = ctx.predict_resident(False)
        t2 = time.perf_counter()
        tm2 = ctx.timers()
        if best is None or (t2 - t0) < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1, tm, tm2)
    tot, tf, tp, tm, tm2 = best
    if ref is None:
        ref = (lml, mu.copy())
    tr = tm["trail_flops"] / (tm["chol_trail"] * 1e-3) / 1e12 if tm["chol_trail"] else 0
    print("%-22s total %.4fs fit %.4f predict %.4f | kbuild %.2f chol %.1f panel %.1f trail %.1f (%.1f TF/s) solve_v %.1f | dlml %.1e dmu %.1e"
          % (cfg, tot, tf, tp, tm["kbuild"], tm["chol"], tm["chol_panel"], tm["chol_trail"], tr, tm2["solve_v"],
             abs(lml - ref[0]) / abs(ref[0]), np.abs(mu - ref[1]).max()), flush=True)
