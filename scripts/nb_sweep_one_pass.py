"""prediction() in one pass (+ alpha) against the outer block width of the Cholesky (GPMI_NB, read at context creation):
   python scripts/nb_sweep_one_pass.py N n nb1 nb2 ...   (0 = the width chosen by size)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N, n = int(sys.argv[1]), int(sys.argv[2])
X, y, Xs = O.synthetic_problem(N, 8, n)
for nb in [int(a) for a in sys.argv[3:]]:
    if nb:
        os.environ["GPMI_NB"] = str(nb)
    else:
        os.environ.pop("GPMI_NB", None)
    with GPContext(0) as ctx:
        ctx.set_train(X, y); ctx.set_test(Xs)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            lml, mu, var = ctx.fit_predict_resident(1.0, 2.0, 5e-4, want_sd=False)
            ctx.alpha()
            ts.append(time.perf_counter() - t0)
        tm = ctx.timers()
        print("N=%d n=%d nb=%4d: one pass + alpha %.2f ms (runs %s)  chol %.2f panel-stream %.2f trail %.2f  lml %.9f" % (
            N, n, nb, min(ts[1:]) * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts[1:]), tm["chol"], tm["chol_panel"], tm["chol_trail"], lml), flush=True)
