"""One-off: the CPU oracle's log-marginal-likelihood for ALL 64 triples of BASELINE config 5 (N=32768, d=8; the
4 x 4 x 4 grid of SURVEY.md section 8d) on the GPU box's host cores, saved as a small fixture.

    gpurun --timeout 1200 -- 'python scripts/oracle_cfg5.py'
    cp gpurun_out/oracle_cfg5_N32768.npz tests/golden/

Each triple is tune_hyperparms_regression.py:306-312 through oracle.fit_predict_feasible (C kernel build with the
reference's per-element arithmetic, LAPACK dpotrf, two triangular solves); four worker processes of 64 BLAS threads
each.  ORACLE-generated (the reference's own kernel build would need 69 GB of temporaries per call)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKERS = int(os.environ.get("ORACLE_CFG5_WORKERS", "4"))


def one(args):
    t, l, sf, s2, N = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gp_oracle as O
    X, y, _ = O.synthetic_problem(N, 8, 4)
    t0 = time.perf_counter()
    lml = O.fit_predict_feasible(X, X[:1], y, sf, l, s2)["lml"]
    return t, float(lml), time.perf_counter() - t0


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    import numpy as np
    import multiprocessing as mp
    triples = np.array([[l, sf, s2] for l in (1., 2., 3., 4.) for sf in (.5, 1., 1.5, 2.) for s2 in (1e-4, 5e-4, 1e-3, 5e-3)])
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    lml = np.full(len(triples), np.nan)
    first = int(os.environ.get("ORACLE_CFG5_FIRST", "0"))       # a call is limited to 20 minutes: the 64 triples take two
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(WORKERS) as pool:
        for t, v, dt in pool.imap_unordered(one, [(t, l, sf, s2, N) for t, (l, sf, s2) in enumerate(triples) if t >= first]):
            lml[t] = v
            print("triple %2d (l=%g sf=%g s2=%g): lml %.9f  (%.1f s; %.0f s so far)" % (t, *triples[t], v, dt, time.perf_counter() - t0), flush=True)
            np.save(os.path.join(out_dir, "oracle_cfg5_N%d_partial.npy" % N), lml)        # survives a kill at the limit
    np.savez(os.path.join(out_dir, "oracle_cfg5_N%d.npz" % N), N=N, d=8, seed=20240531, triples=triples, lml=lml,
             oracle_seconds=time.perf_counter() - t0, host_cores=os.cpu_count(), workers=WORKERS,
             provenance="ORACLE-GENERATED (oracle/gp_oracle.py:fit_predict_feasible per triple on the GPU box's host cores via "
                        "scripts/oracle_cfg5.py); not an output of the reference, which cannot run this size")
    print("done: %d triples in %.0f s" % (len(triples), time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
