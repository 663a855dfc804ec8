"""One-off (CPU, this container): for the triples of BASELINE config 5 whose LML is a cancelling sum (24 and 28 of the
4 x 4 x 4 grid at N=32768: terms of 1e5 .. 5e6 that cancel to 2e3 .. 1.5e4), the magnitude of the terms
|.5 m.m| + |sum log L_ii| + N/2 log 2 pi from the ORACLE (oracle.fit_predict_blocked: tune_hyperparms_regression.py:306-312
with the Cholesky as LAPACK's blocked algorithm at a block of 4096), so that tests/test_parity_gpu.py::test_cfg5_64_triples_N32768 judges those triples against a
scale that does not come from the path under test.  Writes tests/golden/oracle_cfg5_scales.npz (ORACLE-generated)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
N = 32768
which = [int(a) for a in sys.argv[1:]] or [24, 28]
triples = np.array([[l, sf, s2] for l in (1., 2., 3., 4.) for sf in (.5, 1., 1.5, 2.) for s2 in (1e-4, 5e-4, 1e-3, 5e-3)])
X, y, _ = O.synthetic_problem(N, 8, 4)
g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_cfg5_N32768.npz"))
scales, lmls = [], []
for t in which:
    l, sf, s2 = triples[t]
    t0 = time.perf_counter()
    r = O.fit_predict_blocked(X, X[:1], y, sf, l, s2, block=4096)      # LAPACK's dpotrf at N = 32768 segfaults in this container's SciPy build; the blocked form (tied to it in tests/test_oracle_vs_golden.py) does not
    scale = .5 * float(r["m"] @ r["m"]) + abs(float(np.log(r["diagL"]).sum())) + N / 2.0 * np.log(2 * np.pi)
    print("triple %d (l=%g sf=%g s2=%g): lml %.9f (fixture %.9f) term scale %.6e  %.0f s" % (t, l, sf, s2, r["lml"], float(g["lml"][t]), scale, time.perf_counter() - t0), flush=True)
    assert abs(r["lml"] - float(g["lml"][t])) <= 1e-9 * scale
    scales.append(scale); lmls.append(float(r["lml"]))
np.savez(os.path.join(ROOT, "tests", "golden", "oracle_cfg5_scales.npz"), N=N, triple_index=np.array(which), term_scale=np.array(scales),
         lml=np.array(lmls), provenance="ORACLE-GENERATED in the build container by scripts/oracle_cfg5_scales.py "
         "(oracle/gp_oracle.py:fit_predict_blocked); magnitude of the LML's terms for the cancelling triples")
