"""prediction() as two calls (gpmi_factorize + gpmi_predict_resident) against one pass (gpmi_fit_predict_resident: the test
set's rows ride through the Cholesky), alpha included in both: wall per step and the differences of the results.
Usage: python scripts/one_pass_ab.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext

ctx = GPContext(0)
for N in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384, 32768, 65536]:
    n = 1024 if N <= 16384 else 4096
    X, y, Xs = O.synthetic_problem(N, 8, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    reps = 8 if N <= 16384 else 4

    def two():
        lml = ctx.factorize(1.0, 2.0, 5e-4); a = ctx.alpha(); mu, var = ctx.predict_resident(want_sd=False)
        return lml, mu, var, a

    def one():
        lml, mu, var = ctx.fit_predict_resident(1.0, 2.0, 5e-4, want_sd=False); a = ctx.alpha()
        return lml, mu, var, a

    res = {}
    for name, f in (("two", two), ("ride", one), ("follow", one), ("two", two), ("ride", one), ("follow", one)):
        ctx.set_option("one_pass_form", {"two": 0, "ride": 1, "follow": 2}[name])
        f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t0)
        res.setdefault(name, []).append(min(ts))
        res[name + "_out"] = out
        res[name + "_t"] = ctx.timers()
    ctx.set_option("one_pass_form", 0)
    a = res["two_out"]
    fl = N ** 3 / 3 + N * N / 2 + N / 6 + float(N) * N * n
    print("N=%6d n=%5d  two calls %s ms (%.1f TFLOP/s)" % (N, n, ["%.2f" % (t * 1e3) for t in res["two"]], fl / min(res["two"]) / 1e12), flush=True)
    for name in ("ride", "follow"):
        b, t = res[name + "_out"], res[name + "_t"]
        print("        one pass, rows %-6s %s ms (%.1f TFLOP/s)  lml equal %s  max|dmu| %.2e  max|dvar| %.2e  alpha equal %s | chol %.2f (panel %.2f trail %.2f) solve_v %.2f" % (
            name, ["%.2f" % (x * 1e3) for x in res[name]], fl / min(res[name]) / 1e12,
            a[0] == b[0], np.max(np.abs(a[1] - b[1])), np.max(np.abs(a[2] - b[2])), np.array_equal(a[3], b[3]),
            t.get("chol", 0), t.get("chol_panel", 0), t.get("chol_trail", 0), t.get("solve_v", 0)), flush=True)
