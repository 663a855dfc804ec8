"""One-off: the CPU oracle at the HEADLINE size (BASELINE config 3: N=65536, d=8, n=4096, seed 20240531),
run on the GPU box's host cores, saved as a small fixture.

    gpurun --timeout 1200 -- 'python scripts/oracle_fullsize.py --gpu'
    cp gpurun_out/oracle_N65536_d8.npz tests/golden/

What runs is `oracle.gp_oracle.fit_predict_feasible` -- this repo's pinned restatement of
/root/reference/GP_regression.py:138-148 and tune_hyperparms_regression.py:312 (the C kernel build of
rbf_oracle.c, LAPACK dpotrf / dtrsv / dtrsm through SciPy) -- NOT the reference itself, whose broadcast kernel
build needs 275 GB at this size.  The fixture is therefore ORACLE-GENERATED and says so in
its `provenance` field; the oracle's own pinning against the reference's outputs is tests/test_oracle_vs_golden.py.

Stored (< 1 MB): mu (n), var (n), lml, alpha[::8], diagL[::8], m[::8], the oracle's stage times, host core count.
With --gpu the HIP path runs afterwards on the same inputs and the differences are printed and stored in
gpurun_out/oracle_fullsize_diff.json (they are what tests/test_parity_gpu.py's tolerances at this size are set from).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=65536)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--ell", type=float, default=2.0)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--max-seconds", type=float, default=900.0, help="give up early if the factorisation is projected to take longer")
    ap.add_argument("--blocked", type=int, default=0, help="block width of oracle.fit_predict_blocked (0: LAPACK dpotrf on the whole matrix)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import gp_oracle as O
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    out = a.out or os.path.join(out_dir, "oracle_N%d_d%d.npz" % (a.N, a.d))
    X, y, Xs = O.synthetic_problem(a.N, a.d, a.n)
    try:
        import psutil
        print("host: %d cores, %.0f GB RAM available" % (os.cpu_count(), psutil.virtual_memory().available / 1e9), flush=True)
    except Exception:
        print("host: %d cores" % os.cpu_count(), flush=True)
    tm = {}
    t0 = time.perf_counter()
    import threading
    done = threading.Event()

    def heartbeat():                      # gpurun takes 7 silent minutes for a hang
        while not done.wait(60.0):
            print("  ... oracle running, %.0f s, stages so far %s" % (time.perf_counter() - t0, list(tm.keys())), flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    if a.blocked:
        def progress(e, n):
            el = time.perf_counter() - t0
            frac = 1.0 - (1.0 - e / float(n)) ** 3                 # share of the N^3/3 flops behind us
            print("  ... factored %d of %d columns, %.0f s, projected %.0f s" % (e, n, el, el / max(frac, 1e-9)), flush=True)
            if e >= 2 * a.blocked and el / frac > a.max_seconds:
                raise SystemExit("projected factorisation time %.0f s exceeds --max-seconds %.0f" % (el / frac, a.max_seconds))
        ref = O.fit_predict_blocked(X, Xs, y, 1.0, a.ell, 5e-4, block=a.blocked, timings=tm, progress=progress)
    else:
        ref = O.fit_predict_feasible(X, Xs, y, 1.0, a.ell, 5e-4, timings=tm)
    wall = time.perf_counter() - t0
    done.set()
    print("oracle N=%d: %.1f s  stages %s" % (a.N, wall, {k: round(v, 1) for k, v in tm.items()}), flush=True)
    st = a.stride
    np.savez(out, N=a.N, d=a.d, n=a.n, seed=20240531, sigma=1.0, ell=a.ell, noise_var=5e-4, stride=st,
             mu=ref["mu"], var=ref["var"], lml=ref["lml"], alpha_s=ref["alpha"][::st], diagL_s=ref["diagL"][::st],
             m_s=ref["m"][::st], alpha_absmax=np.abs(ref["alpha"]).max(),
             oracle_seconds=wall, oracle_stage_names=np.array(list(tm.keys())), oracle_stage_seconds=np.array(list(tm.values())),
             host_cores=os.cpu_count(),
             provenance="ORACLE-GENERATED (oracle/gp_oracle.py:%s on the GPU box's host cores via "
                        "scripts/oracle_fullsize.py); not an output of the reference, which cannot run this size"
                        % ("fit_predict_blocked, block %d" % a.blocked if a.blocked else "fit_predict_feasible"))
    print("wrote", out, os.path.getsize(out), "bytes", flush=True)
    if not a.gpu:
        return
    from gaussian_process_amd import GPContext
    with GPContext(0) as ctx:
        lml = ctx.fit(X, y, 1.0, a.ell, 5e-4)
        mu, var = ctx.predict(Xs, want_sd=False)
        alpha = ctx.alpha()
        dg = ctx.diag()
        m = ctx.m()
    diff = dict(N=a.N, d=a.d, n=a.n,
                dmu=float(np.max(np.abs(mu - ref["mu"]))), dvar=float(np.max(np.abs(var - ref["var"]))),
                lml_rel=float(abs(lml - ref["lml"]) / abs(ref["lml"])), lml=float(lml), lml_oracle=float(ref["lml"]),
                alpha_rel=float(np.max(np.abs(alpha - ref["alpha"])) / np.max(np.abs(ref["alpha"]))),
                alpha_absmax=float(np.abs(ref["alpha"]).max()),
                diag_rel=float(np.max(np.abs(dg - ref["diagL"]) / ref["diagL"])),
                m_rel=float(np.max(np.abs(m - ref["m"])) / np.max(np.abs(ref["m"]))),
                oracle_seconds=wall, oracle_stages=tm, host_cores=os.cpu_count())
    print(json.dumps(diff), flush=True)
    with open(os.path.join(out_dir, "oracle_fullsize_diff.json"), "w") as f:
        json.dump(diff, f, indent=1)


if __name__ == "__main__":
    main()
