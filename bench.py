#!/usr/bin/env python3
"""Headline benchmark: GP fit + predict at N=65536, d=8, n=4096 test points, fp64.

One "step" = the whole hot path on inputs already resident in HBM:
  K(X,X)+s*I build -> blocked Cholesky (forward solve folded in) -> LML -> backward solve (alpha)
  -> K(X*,X) build -> v = L^-1 K_s sweep -> predictive mean / variance.
value = algorithmic fp64 flops of that path (N^3/3 + N^2/2 + N/6 for the Cholesky,
N^2*n for the triangular solve of K_s) / wall time, whole job, in TFLOP/s.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size N] [--dim d] [--ntest n]

--gpus N > 1: one process per GPU over RCCL.  Under torch.distributed.run (RANK / WORLD_SIZE in the
environment) this process is one rank; started plainly, it spawns the N ranks itself as fresh child
processes BEFORE anything touches the GPU, waits for them and exits with the worst of their codes.
A multi-rank run cannot hang silently: every rank prints a heartbeat line to stderr per phase of every step
("[bench] rank r step k fit"), a watchdog thread in every rank dumps all Python stacks and exits 86 when no
heartbeat came for GPMI_BENCH_STALL_S seconds (default 150), the process group is created with a
GPMI_BENCH_PG_TIMEOUT_S timeout (default 120), and the spawning parent kills every child and exits 124 after
GPMI_BENCH_DEADLINE_S seconds (default 480, under the driver's limit).

--replay-rank r[,r2..] --of G: a REHEARSAL line, not the bench line -- rank r's share of a G-rank run alone on
this one GPU, its collectives served by device copies out of a stored factorisation
(gaussian_process_amd/replay.py); prints the per-rank times and T(1 GPU) / max_r T(replay) = an upper bound of
the G-GPU speed-up.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC (RCCL / cross-process device memory)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# hardware queues per stream-priority level (gaussian_process_amd/_lib.py says why); before anything initialises HIP
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# BASELINE.json's metric, verbatim (it names two quantities: `value` carries the TFLOP/s, `seconds` the time)
BASELINE_METRIC = "GP-fit+predict sec and achieved fp64 TFLOP/s, N=65536 d=8, 1/2/4/8 MI355X"
try:
    BASELINE_METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except (OSError, KeyError, ValueError):
    pass

# nominal MI355X peaks; `peaks_probe` in the output is what this
# box sustains on bare probe kernels in this run (register-only MFMA loop, streaming loads / stores), measured after the
# timed region; every roofline block carries its fraction of both
PEAK_FP64_MFMA_TFLOPS = 78.6     # 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz
PEAK_HBM_GBPS = 8000.0
# north_star targets
TARGET_TRAIL_FRAC = 0.40
TARGET_KBUILD_FRAC = 0.60
TARGET_SPEEDUP_8 = 6.0
TARGET_MEAN_TOL = 1e-8
# the committed rocprofv3 PMC passes `roofline.traffic` is read from (same command, N=65536 n=4096); the file records the
# SHA-256 of the kernel's source at collection time and the line says "stale" when the source has changed since
TRAFFIC_PROFILES = [os.path.join("profiles", f) for f in ("r04_roofline_traffic.json", "r03_roofline_traffic.json",
                                                          "r02c_roofline_traffic.json", "r02_roofline_traffic.json")]
TRAFFIC_KERNEL_SOURCE = os.path.join("gaussian_process_amd", "csrc", "gemm_dma.hip")
# K build: vector instructions per matrix element at d = 8: 35-36 in the interior loop (23 fixed by NumPy's summation
# order + 12 of the exp + the sigma^2 multiply when sigma != 1); counted by rocprofv3 over the build and the K_s build of one
# step: SQ_INSTS_VALU 1.39512e9 wave instructions x 64 lanes / 2.4204e9 elements = 36.9
# (profiles/r03b_pmc_valu_summary.txt; 37.9 before the exponent insertion lost an instruction, r03_pmc_valu_summary.txt), at
# GRBM_GUI_ACTIVE / 8 XCDs / the two launches' ~3.9 ms = 1.9 GHz
KBUILD_VALU_PER_ELEMENT_D8 = 36.9
KBUILD_VALU_PROFILE = os.path.join("profiles", "r03b_kbuild_valu.json")      # the count with the SHA-256 of rbf.hip it was taken on
CPU_HEADLINE_PROFILE = os.path.join("profiles", "cpu_baseline_measured_headline.json")


def kbuild_valu_count():
    """(instructions per element at d = 8, provenance dict): from the committed counter pass; `stale` says whether rbf.hip
    has changed since (as roofline.traffic_source.stale does for the GEMM)"""
    try:
        j = json.load(open(os.path.join(ROOT, KBUILD_VALU_PROFILE)))
        now = source_sha256(j["kernel_source"])
        return float(j["valu_instructions_per_element_d8"]), {
            "file": KBUILD_VALU_PROFILE, "git": j.get("git"), "kernel_source": j["kernel_source"],
            "stale": (now != j.get("kernel_source_sha256")) if now else None,
            "measured": "separate rocprofv3 --pmc SQ_INSTS_VALU pass, not in this run"}
    except (OSError, KeyError, ValueError):
        return KBUILD_VALU_PER_ELEMENT_D8, {"file": None, "stale": None, "measured": "constant in bench.py"}


class Watchdog:
    """Heartbeats to stderr and a thread that ends the process when they stop: a wedged collective must name the rank,
    step and phase it was reached in, not end as an empty stdout at the driver's limit."""

    def __init__(self, rank, stall_s):
        import threading
        self.rank, self.stall_s = rank, float(stall_s)
        self.where, self.t = "start", time.monotonic()
        self.quiet = False
        th = threading.Thread(target=self._watch, daemon=True)
        th.start()

    def beat(self, where):
        self.where, self.t = where, time.monotonic()
        if not self.quiet:
            print("[bench] rank %d %s" % (self.rank, where), file=sys.stderr, flush=True)

    def _watch(self):
        import faulthandler
        while True:
            time.sleep(min(1.0, self.stall_s / 4))
            idle = time.monotonic() - self.t
            if idle > self.stall_s:
                print("[bench watchdog] rank %d: no progress for %.0f s since '%s' -- dumping stacks and exiting 86"
                      % (self.rank, idle, self.where), file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                os._exit(86)          # never re-exec; a plain exit lets the parent reap the other ranks


def source_sha256(rel):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()
    except OSError:
        return None


def probe_peaks(ctx):
    """What this box sustains on bare kernels, right now (a few hundred ms, outside the timed region): the fp64 MFMA
    issue rate with the shader clock it holds meanwhile, and streaming HBM reads / writes over 4 GiB."""
    out = {}
    best, shape = (0.0, 0.0, 0.0), None
    for wps, nacc in ((4, 4), (4, 8), (2, 8), (2, 16)):       # waves per SIMD, independent accumulators per wave
        r = ctx.probe_mfma_f64_ex(wps, nacc, 8192)
        if r[0] > best[0]:
            best, shape = r, (wps, nacc)
    out["fp64_mfma_tflops"], out["mfma_clock_ghz"], out["cycles_per_mfma_per_simd"] = best
    out["mfma_shape"] = {"waves_per_simd": shape[0], "accumulators_per_wave": shape[1]}
    nbytes = 4 << 30
    out["hbm_write_gbps"] = max(ctx.probe_hbm_ex(nbytes, mode, blocks) for mode in (0, 2, 3) for blocks in (4096, 16384))
    out["hbm_read_gbps"] = max(ctx.probe_hbm_ex(nbytes, 4, blocks) for blocks in (2048, 4096, 65536))
    out["note"] = ("register-only v_mfma_f64_16x16x4_f64 loop, one workgroup per CU (best of four shapes; a wave cannot issue "
                   "these back to back at the pipe's rate, so the probe needs 4 waves per SIMD where the GEMM reaches its "
                   "rate with 2) and 16-byte streaming stores / loads over 4 GiB (best of several grid shapes), run after "
                   "the timed steps on the same context")
    return out


def algorithmic_flops(N, n):
    chol = N ** 3 / 3.0 + N ** 2 / 2.0 + N / 6.0
    trsm = float(N) ** 2 * n
    return chol + trsm


def _blas_info():
    threads, desc = os.cpu_count(), "unknown"
    try:
        from threadpoolctl import threadpool_info
        blas = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if blas:
            threads = max(i.get("num_threads") or 1 for i in blas)
            desc = "; ".join("%s %s (%s)" % (i.get("internal_api"), i.get("version"), i.get("threading_layer")) for i in blas)
    except Exception:
        pass
    return threads, desc


def cpu_baseline(d, n_test):
    """BASELINE.md section 3, on this host's cores, bounded to about half a minute of CPU work by default:
      * reference-faithful variant (broadcast (N,d,N) kernel build, np.linalg.cholesky, three LU np.linalg.solve
        calls on the triangular factor: GP_regression.py:18-19,138-144) at N = 2048 and 4096; N = 8192 only with
        GPMI_CPU_BASELINE_FULL=1 (else projected from 4096 and labelled so);
      * memory-feasible variant (the oracle: C kernel build with the reference's per-element arithmetic,
        scipy Cholesky + true triangular solves) at N = 16384 (32768 with GPMI_CPU_BASELINE_FULL=1), stage-timed,
        as what it is: a sample.
    `value` is the measured feasible-variant rate of this run's sample; `measured_headline` is the same oracle timed once
    at the headline N on a GPU box's host (profiles/cpu_baseline_measured_headline.json), quoted with its provenance."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gp_oracle as O
    lib = os.path.join(ROOT, "oracle", "build", "librbf_oracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    full = os.environ.get("GPMI_CPU_BASELINE_FULL") == "1"
    ell = 2.0 * np.sqrt(d / 8.0)
    threads, blas = _blas_info()

    faithful = []
    nf = 1024
    for Nf in (2048, 4096, 8192):
        if Nf == 8192 and not full:
            t4 = faithful[-1]["seconds"]
            faithful.append({"N": Nf, "n_test": nf, "seconds": None, "projected_seconds": t4 * 4.7,
                             "note": "not run in the default (bounded) bench: projected from N=4096 with the "
                                     "N=4096 -> 8192 ratio measured in BASELINE.md section 2 (42.8 / 9.12); "
                                     "GPMI_CPU_BASELINE_FULL=1 runs it"})
            continue
        Xf, yf, Xsf = O.synthetic_problem(Nf, d, nf)
        t1 = time.perf_counter()
        O.posterior(Xf, Xsf, yf, 1.0, ell, 5e-4)
        faithful.append({"N": Nf, "n_test": nf, "seconds": time.perf_counter() - t1})

    Ns = int(os.environ.get("GPMI_CPU_BASELINE_N", "32768" if full else "16384"))
    X, y, Xs = O.synthetic_problem(Ns, d, n_test)
    stages = {}
    t0 = time.perf_counter()
    O.fit_predict_feasible(X, Xs, y, 1.0, ell, 5e-4, timings=stages)
    dt = time.perf_counter() - t0

    # the headline size itself, MEASURED (one run of the same oracle on a GPU box's host, committed with its log): the
    # N^3 extrapolation of the N = 16384 stage times that stood here until round 3 read 569 s where the measurement
    # says 199 s -- dpotrf runs at 0.19 TFLOP/s at N = 16384 and at 0.59 at N = 65536 -- and is gone
    measured = None
    try:
        measured = json.load(open(os.path.join(ROOT, CPU_HEADLINE_PROFILE)))
        measured["tflops"] = algorithmic_flops(measured["N"], measured["n_test"]) / measured["seconds"] / 1e12
        measured["label"] = "measured, not in this run"
        measured["sample_rate_over_headline_rate"] = (algorithmic_flops(Ns, n_test) / dt) / \
            (algorithmic_flops(measured["N"], measured["n_test"]) / measured["seconds"])
    except (OSError, KeyError, ValueError):
        pass
    return {"value": algorithmic_flops(Ns, n_test) / dt / 1e12, "unit": "TFLOP/s", "cores": threads,
            "kind": "port", "seconds": dt, "blas": blas, "os_cpu_count": os.cpu_count(),
            "stages_s": stages,
            "reference_faithful": faithful,
            "reference_faithful_note": "broadcast RBF + np.linalg.cholesky + 3 LU np.linalg.solve, as the reference "
                                       "issues them; d=%d, n_test=%d; cannot run beyond N~8192 (O(N^2 d) temporaries)" % (d, nf),
            "measured_headline": measured,
            "sample": "oracle (memory-feasible restatement) fit+predict at N=%d d=%d n=%d, same generator and "
                      "hyper-parameters as the GPU run" % (Ns, d, n_test)}


def spawn_ranks(n):
    """Parent of a plain `python bench.py --gpus n`: start the n ranks as fresh children (the parent never touches
    the GPU), wait, return the worst exit code.  A failed rank takes the others down after a grace period."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = float(os.environ.get("GPMI_BENCH_DEADLINE_S", "480"))
    grace = float(os.environ.get("GPMI_BENCH_GRACE_S", "30"))
    t_start = time.monotonic()
    worst, failed_at, timed_out = 0, None, False
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0) and failed_at is None:
                failed_at = time.monotonic()
                print("[bench parent] rank %d exited with code %s; the others get %.0f s" % (r, rc, grace),
                      file=sys.stderr, flush=True)
        over = time.monotonic() - t_start > deadline
        if over and not timed_out:
            timed_out = True
            alive = [r for r, p in enumerate(procs) if p.poll() is None]
            print("[bench parent] deadline of %.0f s reached with ranks %s still running: killing them (their last "
                  "heartbeat lines above say where)" % (deadline, alive), file=sys.stderr, flush=True)
        if over or (failed_at is not None and time.monotonic() - failed_at > grace):
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = p.wait()
        if rc != 0:
            worst = max(worst, rc if rc > 0 else 1)
    if timed_out:
        worst = 124
    return worst


def rehearse_cpu(args):
    """The multi-rank bench's skeleton on gloo and CPU tensors (GPMI_BENCH_REHEARSE=gloo_cpu; tests/test_dist.py): the same
    watchdog, heartbeats, process-group timeout and parent deadline, one all-reduce per phase.  GPMI_BENCH_STALL_RANK /
    GPMI_BENCH_STALL_STEP make one rank sleep in front of a collective, as a wedged rank would."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    sys.stdout.flush()
    stdout_fd = os.dup(1)                 # as in main(): stdout carries the one result line, library chatter goes to stderr
    os.dup2(2, 1)
    wd = Watchdog(rank, os.environ.get("GPMI_BENCH_STALL_S", "150"))
    wd.beat("init_process_group gloo")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("GPMI_BENCH_PG_TIMEOUT_S", "120"))))
    wd.beat("process group up")
    stall_rank = int(os.environ.get("GPMI_BENCH_STALL_RANK", "-1"))
    stall_step = int(os.environ.get("GPMI_BENCH_STALL_STEP", "0"))
    t = torch.ones(4, dtype=torch.float64)
    for k in range(args.warmup + args.steps):
        for phase in ("fit", "alpha", "predict"):
            wd.beat("step %d %s" % (k, phase))
            if rank == stall_rank and k == stall_step and phase == "alpha":
                time.sleep(1e6)
            dist.all_reduce(t)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps({"rehearsal": "gloo_cpu", "ranks": world, "steps": args.steps}), flush=True)
        os.dup2(2, 1)
    dist.destroy_process_group()
    return 0


def extra_configs(ctx, wd):
    """The other BASELINE configs under the same clock as the headline (after its timed steps, same context): config 2
    (N=16384, d=8, n=1024), config 5's 64 triples at N=32768 on this one GPU, and config 4's workload (N=131072, d=16)
    on one GPU -- wall seconds and achieved TFLOP/s each; ~45 s together."""
    import numpy as np
    out = {}
    rng = np.random.default_rng(20240531)

    def problem(N, d, n):
        X = rng.uniform(-1, 1, (N, d))
        y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N)
        return X, y, rng.uniform(-1, 1, (n, d))

    one_pass = os.environ.get("GPMI_BENCH_TWO_CALLS") != "1"          # the same call form as the headline's step

    def one_step(ell, two_calls=False):
        if one_pass and not two_calls:
            lml, mu, var = ctx.fit_predict_resident(1.0, ell, 5e-4, want_sd=False)
            alpha = ctx.alpha()
            return lml, mu
        lml = ctx.factorize(1.0, ell, 5e-4)
        alpha = ctx.alpha()
        mu, var = ctx.predict_resident(want_sd=False)
        return lml, mu

    # config 2
    wd.beat("extra config 2 (N=16384)")
    N2, n2 = 16384, 1024
    X, y, Xs = problem(N2, 8, n2)
    ctx.set_train(X, y)
    ctx.set_test(Xs)
    one_step(2.0)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lml, mu = one_step(2.0)
    dt = (time.perf_counter() - t0) / reps
    fl = algorithmic_flops(N2, n2)
    out["cfg2_N16384_d8_n1024"] = {"ms_per_step": dt * 1e3, "tflops": fl / dt / 1e12, "frac_of_fp64_mfma_peak": fl / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                                   "steps": reps, "lml": float(lml), "finite": bool(np.all(np.isfinite(mu)))}
    if one_pass:
        one_step(2.0, True)
        t0 = time.perf_counter()
        for _ in range(reps):
            one_step(2.0, True)
        out["cfg2_N16384_d8_n1024"]["ms_per_step_two_calls"] = (time.perf_counter() - t0) / reps * 1e3
    # config 5 on one GPU
    wd.beat("extra config 5 (64 triples, N=32768)")
    N5 = 32768
    X, y, _ = problem(N5, 8, 4)
    triples = np.array([[l, sf, s2] for l in (1., 2., 3., 4.) for sf in (.5, 1., 1.5, 2.) for s2 in (1e-4, 5e-4, 1e-3, 5e-3)])
    ctx.set_train(X, y)
    ctx.lml_batch(triples[:2])
    wd.stall_s = max(wd.stall_s, 300.0)
    t0 = time.perf_counter()
    lmls, st = ctx.lml_batch(triples)
    dt = time.perf_counter() - t0
    out["cfg5_64_triples_N32768_one_gpu"] = {"seconds": dt, "seconds_per_triple": dt / len(triples),
                                             "tflops": len(triples) * (N5 ** 3 / 3.0) / dt / 1e12,
                                             "frac_of_fp64_mfma_peak": len(triples) * (N5 ** 3 / 3.0) / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                                             "failed": int(np.sum(st)), "finite": bool(np.all(np.isfinite(lmls)))}
    # config 4's workload on one GPU
    wd.beat("extra config 4 workload (N=131072, d=16) on one GPU")
    N4, d4, n4 = 131072, 16, 4096
    X, y, Xs = problem(N4, d4, n4)
    ctx.set_train(X, y)
    ctx.set_test(Xs)
    t0 = time.perf_counter()
    one_step(2.8)                        # first touch of 138 GB: reported as `seconds_first_step`
    dt_cold = time.perf_counter() - t0
    wd.beat("extra config 4 workload, second step")
    t0 = time.perf_counter()
    lml, mu = one_step(2.8)
    dt = time.perf_counter() - t0
    fl = algorithmic_flops(N4, n4)
    out["cfg4_workload_N131072_d16_one_gpu"] = {"seconds": dt, "seconds_first_step": dt_cold, "tflops": fl / dt / 1e12,
                                                "frac_of_fp64_mfma_peak": fl / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS, "steps": 1,
                                                "lml": float(lml), "finite": bool(np.all(np.isfinite(mu))),
                                                "note": "config 4 names 8 GPUs; this is its problem on ONE (138 GB of the 288 GB): the second of two steps (the first touches the allocation for the first time)"}
    out["call_form"] = "one pass (gpmi_fit_predict_resident + gpmi_get_alpha)" if one_pass else "two calls"
    out["note"] = ("timed in this run after the headline's steps, same context and generator; TFLOP/s on algorithmic flops "
                   "(N^3/3 + N^2/2 + N/6 + N^2 n; config 5: N^3/3 per triple)")
    return out


def run_replay(args):
    """--replay-rank r[,r2...] --of G: each named rank's share of a G-rank run alone on this GPU (replay.py), timed as the
    bench times a step; T(1 GPU, this run) / max_r T(replay) bounds the G-GPU speed-up from above."""
    import numpy as np
    import torch
    from gaussian_process_amd import GPContext
    from gaussian_process_amd.replay import ReplaySource, replay_rank, check_rank
    N, d, n, G = args.size, args.dim, args.ntest, args.of
    ranks = [int(r) for r in str(args.replay_rank).split(",")]
    ell, sigma, s = 2.0 * np.sqrt(d / 8.0), 1.0, 5e-4
    rng = np.random.default_rng(20240531)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N)
    Xs = rng.uniform(-1, 1, (n, d))
    torch.cuda.set_device(0)
    wd = Watchdog(0, os.environ.get("GPMI_BENCH_STALL_S", "300"))
    nb_auto = 256
    while nb_auto < 2048 and N // (2 * nb_auto) >= 4 * G:
        nb_auto *= 2
    nb = int(os.environ.get("GPMI_DIST_NB", str(nb_auto)))
    lookahead = int(os.environ.get("GPMI_DIST_LOOKAHEAD", "2"))
    layout = os.environ.get("GPMI_DIST_LAYOUT", "balanced")
    one_pass = os.environ.get("GPMI_BENCH_TWO_CALLS") != "1"      # the bench line's call form (prediction() in one pass)
    # T(1 GPU): the product's single-GPU path, same step as the bench line
    wd.beat("single-GPU reference")
    t1_ms = None
    if os.environ.get("GPMI_REPLAY_NO_T1") != "1":
        with GPContext(0) as ctx:
            ctx.set_train(X, y)
            ctx.set_test(Xs)
            for k in range(args.warmup + args.steps):
                if k == args.warmup:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                if one_pass:
                    lml1, _, _ = ctx.fit_predict_resident(sigma, ell, s, want_sd=False)
                    ctx.alpha()
                else:
                    lml1 = ctx.factorize(sigma, ell, s)
                    ctx.alpha()
                    ctx.predict_resident(want_sd=False)
            torch.cuda.synchronize()
            t1_ms = (time.perf_counter() - t0) / args.steps * 1e3
    wd.beat("source factorisation (block rows %d)" % nb)
    src = ReplaySource(0, nb, X, y, Xs, sigma, ell, s, lookahead=lookahead)
    res = []
    for r in ranks:
        wd.beat("replay rank %d of %d" % (r, G))
        gp = replay_rank(0, src, r, G, X, y, Xs, lookahead=lookahead, layout=layout)

        def step():
            if one_pass:
                lml, mu, var = gp.fit_predict_resident(sigma, ell, s, want_sd=False)
                alpha = gp.alpha()
                return lml, mu, var, alpha
            lml = gp.factorize(sigma, ell, s)
            alpha = gp.alpha()
            mu, var = gp.predict_resident(want_sd=False)
            return lml, mu, var, alpha
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lml, mu, var, alpha = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        tm = gp.timers()
        chk = check_rank(gp, src)
        one = {"rank": r, "ms_per_step": ms, "fit_ms": tm.get("fit"), "alpha_ms": tm.get("alpha"), "predict_ms": tm.get("predict"),
               "row_blocks": gp.nloc, "carries_y": gp.yrow is not None,
               "lml_rel_vs_source": abs(lml - src.lml) / abs(src.lml),
               "mu_maxabs_vs_source": float(np.max(np.abs(mu - src.mu))), "var_maxabs_vs_source": float(np.max(np.abs(var - src.var))),
               "alpha_rel_vs_source": float(np.max(np.abs(alpha - src.alpha_h)) / np.max(np.abs(src.alpha_h))),
               "delivered_bytes_per_step": {k: v // (args.warmup + args.steps) for k, v in gp.comm.bytes.items()}}
        one.update(chk)
        gp.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        diag = {"step_wall_ms": (time.perf_counter() - t0) * 1e3}
        for key, v in gp.profile_summary().items():
            diag[key + "_ms"] = round(v["ms"], 3)
            diag[key + "_n"] = v["n"]
            if key in ("stall_panel", "host_issue", "allgather", "bcast"):
                diag[key + "_max_ms"] = round(v["max_ms"], 3)
        gp.profile(False)
        one["diag"] = diag
        res.append(one)
        del gp
        torch.cuda.empty_cache()
    worst = max(o["ms_per_step"] for o in res)
    return {"replay": True, "what": "one rank of a G-rank DistGP run alone on one GPU: its row blocks, panel solves, update launches, "
            "streams, events, pack copies and Python issue are a real rank's; collectives are device copies out of a stored "
            "factorisation (gaussian_process_amd/replay.py), so xGMI time is NOT in these numbers",
            "of": G, "ranks": res, "block_rows": nb, "lookahead": lookahead, "layout": layout,
            "call_form": "one pass (fit_predict_resident + alpha)" if one_pass else "two calls (factorize + alpha + predict_resident)",
            "config": {"workload": "GP fit+predict N=%d d=%d n_test=%d" % (N, d, n), "N": N, "d": d, "n_test": n},
            "t1_ms": t1_ms, "t1_what": "single-GPU path (GPContext), same step, this run",
            "worst_rank_ms": worst,
            "speedup_upper_bound": (t1_ms / worst) if t1_ms else None,
            "steps": args.steps, "warmup": args.warmup, "lml_source": src.lml}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--ntest", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-form", action="store_true",
                    help="do not time the other call form after the timed region (profiling passes: every launch of the run "
                         "then belongs to the timed form)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the other BASELINE configs (2, 5 and 4's workload on one GPU) behind the headline steps")
    ap.add_argument("--replay-rank", default=None,
                    help="rehearsal: comma-separated ranks of a --of G run to replay alone on this GPU")
    ap.add_argument("--of", type=int, default=8, help="world size of the replayed run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))      # before torch / HIP are imported in this process
    echo = os.environ.get("GPMI_BENCH_SPAWN_ECHO")
    if echo:                                  # tests/test_dist.py: what a spawned rank sees, no GPU needed
        print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "addr": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT")}), flush=True)
        sys.exit(3 if echo == "fail" + os.environ.get("RANK", "0") else 0)
    if os.environ.get("GPMI_BENCH_REHEARSE") == "gloo_cpu":
        sys.exit(rehearse_cpu(args))      # tests/test_dist.py: the watchdog and the parent's deadline, no GPU needed

    # stdout carries exactly ONE line, the result: whatever libraries print meanwhile (RCCL's version banner under
    # NCCL_DEBUG=VERSION goes to stdout at communicator creation) is sent to stderr by pointing fd 1 there until the end
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.replay_rank is not None:
        out = run_replay(args)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        return
    wd = Watchdog(rank, os.environ.get("GPMI_BENCH_STALL_S", "150"))
    wd.quiet = world == 1 and os.environ.get("GPMI_BENCH_FORCE_DIST") != "1"      # one rank: the thread still watches
    N, d, n = args.size, args.dim, args.ntest
    ell, sigma, s = 2.0 * np.sqrt(d / 8.0), 1.0, 5e-4

    rng = np.random.default_rng(20240531)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N)
    Xs = rng.uniform(-1, 1, (n, d))

    # rehearsal on a one-GPU box (tests only): GPMI_BENCH_BACKEND=gloo GPMI_BENCH_ONE_DEVICE=1
    if os.environ.get("GPMI_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GPMI_BENCH_BACKEND", "nccl")
    # GPMI_DIST_COMM=rccl: the data path's collectives go through this library's own RCCL binding (gpmi_comm_*); the process
    # group is then only the control plane (id exchange, barriers, the result gather) and runs on gloo
    if os.environ.get("GPMI_DIST_COMM", "torch") == "rccl" and backend == "nccl":
        backend = "gloo"
    torch.cuda.set_device(local_rank)
    # GPMI_BENCH_FORCE_DIST=1: run the multi-rank driver (and every RCCL collective it issues) on a world
    # of one rank -- the distributed code path at full size on a single GPU (rehearsal, not a bench line)
    force_dist = os.environ.get("GPMI_BENCH_FORCE_DIST") == "1"
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    multi = world > 1 or force_dist
    if multi:
        import datetime
        import torch.distributed as dist
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("GPMI_BENCH_PG_TIMEOUT_S", "120")))
        wd.beat("init_process_group %s" % backend)
        if backend == "nccl":
            # the process group's internal communication stream at high priority: its kernels (a few workgroups per channel)
            # are placed ahead of the update's next tiles instead of taking turns with them
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("GPMI_BENCH_NCCL_HIGH_PRIORITY", "1") == "1")
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout, pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        wd.beat("process group up")
        from gaussian_process_amd.dist import DistGP
        # block rows: as large as leaves every rank >= 4 blocks, capped at 2048 -- the update GEMM is more efficient at
        # larger depth and there are half the steps to issue; with the balanced dealing 4 blocks per rank still put the
        # worst rank within 2 % of the mean share (8-rank replay, every rank: 244.4 ms worst at 2048, 247.5 at 1024:
        # profiles/r04_replay_onepass_G8_nb2048.json)
        nb_auto = 256
        while nb_auto < 2048 and N // (2 * nb_auto) >= 4 * max(world, 1):
            nb_auto *= 2
        nb_used = int(os.environ.get("GPMI_DIST_NB", str(nb_auto)))
        gp = DistGP(local_rank, nb=nb_used,
                    lookahead=int(os.environ.get("GPMI_DIST_LOOKAHEAD", "2")), force_collectives=force_dist)
        gp.set_train(X, y)
        gp.set_test(Xs)

        wd.beat("train / test sets resident")

        dist_one_pass = os.environ.get("GPMI_BENCH_TWO_CALLS") != "1"

        def step(k=-1):
            if dist_one_pass:       # prediction() in one pass: every rank's share of the test rows rides below its blocks
                wd.beat("step %d fit + predict" % k)
                lml, mu, var = gp.fit_predict_resident(sigma, ell, s, want_sd=False)
                wd.beat("step %d alpha" % k)
                alpha = gp.alpha()
                return lml, mu, var, alpha
            wd.beat("step %d fit" % k)
            lml = gp.factorize(sigma, ell, s)
            wd.beat("step %d alpha" % k)
            alpha = gp.alpha()
            wd.beat("step %d predict" % k)
            mu, var = gp.predict_resident(want_sd=False)
            return lml, mu, var, alpha

        def barrier():
            wd.beat("barrier")
            dist.barrier()
            torch.cuda.synchronize()
        timers_fn = gp.timers
    else:
        from gaussian_process_amd import GPContext
        ctx = GPContext(local_rank)
        ctx.set_train(X, y)      # inputs resident in HBM before the timed region
        ctx.set_test(Xs)

        # prediction() has both sets up front (GP_regression.py:109), so the step is the one-pass form: the test set's rows go
        # through the Cholesky with the training rows (gpmi_fit_predict_resident), then alpha.  GPMI_BENCH_TWO_CALLS=1 times
        # gpmi_factorize + gpmi_get_alpha + gpmi_predict_resident instead; the other form is timed after the timed region
        # either way and reported beside it
        one_pass = os.environ.get("GPMI_BENCH_TWO_CALLS") != "1"

        def step_two(k=-1):
            wd.beat("step %d" % k)
            lml = ctx.factorize(sigma, ell, s)
            alpha = ctx.alpha()
            mu, var = ctx.predict_resident(want_sd=False)
            return lml, mu, var, alpha

        def step_one(k=-1):
            wd.beat("step %d" % k)
            lml, mu, var = ctx.fit_predict_resident(sigma, ell, s, want_sd=False)
            alpha = ctx.alpha()
            return lml, mu, var, alpha

        step = step_one if one_pass else step_two

        def barrier():
            torch.cuda.synchronize()
        timers_fn = None

    for k in range(args.warmup):
        step(k - args.warmup)
    # Multi-rank: which form of a rank's large update launches?  One workgroup per tile loses 12 % for every shader engine in
    # which another kernel holds a CU exclusively (LAB_NOTES.md: per-tile launches are dealt statically down to the engine);
    # the ticket form loses only the CU's share but slows the panel solves that share CUs with it.  Whether RCCL's kernels
    # hold CUs exclusively beside an update workgroup cannot be known on a one-GPU box, so it is MEASURED here, before the
    # timed region: two untimed steps in each form, the maximum over ranks decides, every rank takes the same form.
    update_form = None
    if multi and world > 1 and os.environ.get("GPMI_DIST_TICKET", "auto") == "auto" and backend != "gloo":
        trial = {}
        for form in (0, 1):
            gp.ticket = form
            step(-1)
            barrier()
            tt = time.perf_counter()
            step(-1)
            step(-1)
            barrier()
            t = torch.tensor([(time.perf_counter() - tt) / 2], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trial[form] = float(t.item())
        gp.ticket = 1 if trial[1] < 0.98 * trial[0] else 0
        update_form = {"chosen": "ticket" if gp.ticket else "per-tile", "per_tile_ms": trial[0] * 1e3, "ticket_ms": trial[1] * 1e3,
                       "how": "two untimed steps in each form before the timed region, maximum over ranks; ticket needs 2 % to win"}
        wd.beat("update form: %s" % update_form["chosen"])
    stage = {}
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        lml, mu, var, alpha = step(k)
        if not multi:
            tf = ctx.timers()
            for k, v in tf.items():
                stage[k] = stage.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if multi:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        stage = timers_fn()
        mine = torch.tensor([stage.get("fit", 0.0), stage.get("alpha", 0.0), stage.get("predict", 0.0)],
                            device="cuda", dtype=torch.float64)
        allr = torch.empty(world * 3, device="cuda", dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        per_rank = [{"rank": r, "fit_ms": float(v[0]), "alpha_ms": float(v[1]), "predict_ms": float(v[2])}
                    for r, v in enumerate(allr.view(world, 3).cpu().numpy())]
        # one more step OUTSIDE the timed region with the schedule's self-diagnosis on (DistGP.profile): per rank and
        # per kind, event-timed on the stream each piece runs on -- update launches, all-gathers, broadcasts, the main
        # stream's wait for each panel (= what the panel chain costs the MFMA stream) -- and the host's issue time per
        # step.  A sub-target scaling number can then be read: compute (update_ms against 1/world of the single-GPU
        # update time), exposed communication (stall_panel_ms), or a host that cannot issue fast enough (host_issue_ms
        # against the step's wall).
        diag_all = None
        if os.environ.get("GPMI_BENCH_NO_DIAG") != "1":
            gp.profile(True)
            barrier()
            t1 = time.perf_counter()
            step(args.steps)
            barrier()
            diag = {"rank": rank, "step_wall_ms": (time.perf_counter() - t1) * 1e3}
            for key, v in gp.profile_summary().items():
                diag[key + "_ms"] = round(v["ms"], 3)
                diag[key + "_n"] = v["n"]
                if key in ("stall_panel", "host_issue", "allgather", "bcast"):
                    diag[key + "_max_ms"] = round(v["max_ms"], 3)
            gp.profile(False)
            diag_all = [None] * world
            dist.all_gather_object(diag_all, diag)
    ms_per_step = dt / args.steps * 1e3
    flops = algorithmic_flops(N, n)
    value = flops / (dt / args.steps) / 1e12
    other_form = None
    if not multi and not args.no_other_form:
        # the other call form, outside the timed region: its wall, its results against the timed form's, and -- from the
        # two-call form, where a7 runs alone -- the stage timers of the predict sweep
        other = step_two if one_pass else step_one
        other(-1)
        nrep = max(1, min(args.steps, 3))
        torch.cuda.synchronize()
        tt = time.perf_counter()
        for _ in range(nrep):
            o_lml, o_mu, o_var, o_alpha = other(-1)
        torch.cuda.synchronize()
        o_ms = (time.perf_counter() - tt) / nrep * 1e3
        if one_pass:
            tf = ctx.timers()
            for kk in ("ks", "solve_v", "meanvar"):
                stage[kk] = tf.get(kk, 0.0) * args.steps          # a7 alone (two-call form), scaled like the summed timers
        other_form = {"form": "two calls: gpmi_factorize + gpmi_get_alpha + gpmi_predict_resident" if one_pass
                      else "one pass: gpmi_fit_predict_resident + gpmi_get_alpha",
                      "ms_per_step": o_ms, "steps": nrep, "tflops": flops / (o_ms * 1e-3) / 1e12,
                      "lml_equal": bool(o_lml == lml), "max_abs_dmu": float(np.max(np.abs(o_mu - mu))),
                      "max_abs_dvar": float(np.max(np.abs(o_var - var))), "alpha_equal": bool(np.array_equal(o_alpha, alpha)),
                      "note": "timed after the timed region, same context and inputs"}

    if rank == 0:
        assert np.all(np.isfinite(mu)) and np.isfinite(lml) and np.all(np.isfinite(alpha))
        out = {
            "metric": BASELINE_METRIC,
            "value": value, "unit": "TFLOP/s", "seconds": dt / args.steps,
            "value_is": "achieved fp64 TFLOP/s of the whole fit+predict job (algorithmic flops / wall); "
                        "the metric's seconds are in `seconds`",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "GP fit+predict N=%d d=%d n_test=%d (BASELINE configs[2])" % (N, d, n),
                       "N": N, "d": d, "n_test": n, "ell": float(ell), "sigma_f": sigma, "noise_var": s,
                       "partition": "single GPU" if world == 1 else "row-block cyclic x%d" % world},
            "lml": float(lml),
        }
        if not multi:
            out["config"]["call_form"] = ("one pass: gpmi_fit_predict_resident (K build, Cholesky with the test set's rows "
                                          "K(X*, X) carried along = a7, LML, mean, variance) + gpmi_get_alpha" if one_pass
                                          else "two calls: gpmi_factorize + gpmi_get_alpha + gpmi_predict_resident")
            out["other_call_form"] = other_form
        targets = {}
        if multi:
            out["rccl_ranks"] = dist.get_world_size()
            out["backend"] = backend
            out["config"]["block_rows"] = nb_used
            out["config"]["row_block_layout"] = gp.layout
            out["config"]["call_form"] = ("one pass: DistGP.fit_predict_resident (each rank's share of the test rows rides below "
                                          "its row blocks; no message of their own) + alpha" if dist_one_pass
                                          else "two calls: factorize + alpha + predict_resident")
            out["config"]["update_form"] = update_form if update_form is not None else ("ticket" if gp.ticket else "per-tile")
            out["stages_ms"] = stage            # last step, rank 0: fit / alpha / predict wall
            out["per_rank_ms"] = per_rank
            out["per_rank_diag"] = diag_all
            out["per_rank_diag_note"] = ("one extra step outside the timed region, events on the stream each piece runs on: "
                                         "update = trailing-update launches of the fit, allgather / bcast / pack = panel "
                                         "collectives and the pack copy, stall_panel = main stream waiting for the panel "
                                         "chain, diag / panel_solve = diagonal-block factorisations and panel solves, "
                                         "*_v = predict sweep, alpha_* = backward solve, host_issue = host time to issue "
                                         "each fit step")
            out["comm"] = gp.comm.describe()
            if force_dist:
                out["config"]["partition"] = "multi-rank driver forced on one rank (RCCL communicator of size 1)"
            targets["speedup_8gpu"] = {"target": TARGET_SPEEDUP_8, "note": "value at n_gpus=8 / value at n_gpus=1; "
                                       "computed by the driver from the per-N lines"}
        if not multi and stage:
            k = args.steps
            trail_ms = stage.get("chol_trail", 0.0) / k
            trail_flops = stage.get("trail_flops", 0.0) / k
            launches = stage.get("trail_launches", 0.0) / k
            ach = trail_flops / (trail_ms * 1e-3) / 1e12 if trail_ms > 0 else 0.0
            traffic, traffic_src = None, None
            if N == 65536 and n == 4096:
                # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same
                # command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; scripts/rocpd_extract.py traffic)
                for rel in TRAFFIC_PROFILES:
                    tpath = os.path.join(ROOT, rel)
                    if os.path.exists(tpath):
                        tj = json.load(open(tpath))
                        traffic = tj.get("traffic_bytes_per_launch")
                        sha_then, sha_now = tj.get("kernel_source_sha256"), source_sha256(TRAFFIC_KERNEL_SOURCE)
                        traffic_src = {"file": rel, "git": tj.get("git"), "measured": "separate rocprofv3 --pmc passes, "
                                       "not in this run", "kernel_source": TRAFFIC_KERNEL_SOURCE,
                                       "stale": (sha_then != sha_now) if sha_then and sha_now else None,
                                       "stale_means": "the kernel's source file has changed since the counters were "
                                                      "collected (SHA-256 recorded in the profile); null: unknown"}
                        break
            out["roofline"] = {
                "kernel": "chol_trailing_update_dma_kernel (Cholesky trailing update: 128x128 tile, 8 waves x 2x4 "
                          "v_mfma_f64_16x16x4_f64, LDS-DMA staging)",
                "bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src,
                "launches_per_step": launches,
                "flops_per_launch": trail_flops / launches if launches else 0.0,
                "avg_launch_ms": trail_ms / launches if launches else 0.0}
            out["stages_ms"] = {kk: vv / k for kk, vv in stage.items() if not kk.startswith("trail_")}
            targets["trailing_update_mfma_frac"] = {"target": TARGET_TRAIL_FRAC, "achieved": ach / PEAK_FP64_MFMA_TFLOPS,
                                                    "met": ach / PEAK_FP64_MFMA_TFLOPS >= TARGET_TRAIL_FRAC}
            kb = stage.get("kbuild", 0.0) / k
            if kb > 0:
                Np = (N + 127) // 128 * 128
                T = Np // 128
                kbytes = 8.0 * 128 * 128 * T * (T + 1) / 2 + 16.0 * N * d
                frac = kbytes / (kb * 1e-3) / 1e9 / PEAK_HBM_GBPS
                out["kbuild_hbm"] = {"bound": "hbm", "achieved": kbytes / (kb * 1e-3) / 1e9,
                                     "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": frac,
                                     "bytes": kbytes, "note": "lower tiles incl. diagonal"}
                targets["kbuild_hbm_frac"] = {"target": TARGET_KBUILD_FRAC, "achieved": frac, "met": frac >= TARGET_KBUILD_FRAC}
                if d == 8:
                    # the kernel is bound by vector-instruction issue, not by HBM: ~37 fp64-rate instructions per element
                    # (23 of them NumPy's summation order, which bit-exact parity fixes) on 1024 SIMDs of 16 lanes
                    KBUILD_VALU, valu_src = kbuild_valu_count()
                    out["kbuild_hbm"]["valu_count_source"] = valu_src
                    elems = 128.0 * 128 * T * (T + 1) / 2
                    for ghz in (2.4, 1.9):
                        ceil_gbps = 1024 * 16 * ghz * 1e9 / KBUILD_VALU * 8 / 1e9
                        out["kbuild_hbm"]["valu_ceiling_gbps_at_%.1fGHz" % ghz] = ceil_gbps
                    out["kbuild_hbm"]["valu_instructions_per_element"] = KBUILD_VALU
                    out["kbuild_hbm"]["valu_issue_ghz_implied"] = elems * KBUILD_VALU / (1024 * 16) / (kb * 1e-3) / 1e9
                    out["kbuild_hbm"]["bound_note"] = ("VALU-issue bound: elements x 36.9 counted instructions / (1024 SIMDs x 16 lanes) / time "
                                                       "= the shader clock the kernel would need if it did nothing but issue them "
                                                       "(valu_issue_ghz_implied); the counters put the clock it holds on this fp64 + store mix at 1.9 GHz, so "
                                                       "the kernel runs at ~3/4 of what instruction issue alone allows and the 0.60 target sits at 0.73 of it")
                    targets["kbuild_valu_ceiling"] = {"ceiling_gbps_at_1.9GHz": 1024 * 16 * 1.9e9 / KBUILD_VALU * 8 / 1e9,
                                                      "achieved_gbps": kbytes / (kb * 1e-3) / 1e9,
                                                      "achieved_of_ceiling": kbytes / (kb * 1e-3) / 1e9 / (1024 * 16 * 1.9e9 / KBUILD_VALU * 8 / 1e9)}
            al = stage.get("alpha", 0.0) / k
            if al > 0:
                abytes = 8.0 * N * (N + 1) / 2
                out["alpha_hbm"] = {"bound": "hbm", "achieved": abytes / (al * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS,
                                    "unit": "GB/s", "frac": abytes / (al * 1e-3) / 1e9 / PEAK_HBM_GBPS, "bytes": abytes,
                                    "ms": al, "note": "a5: backward solve L^T alpha = m reads the triangle once"}
            sv = stage.get("solve_v", 0.0) / k
            if sv > 0:
                vflops = float(N) * N * n
                out["solve_v_mfma"] = {"bound": "mfma", "achieved": vflops / (sv * 1e-3) / 1e12, "peak": PEAK_FP64_MFMA_TFLOPS,
                                       "unit": "TFLOP/s", "frac": vflops / (sv * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                                       "flops": vflops, "ms": sv,
                                       "note": "a7: v = L^-1 K_s as a blocked sweep (trsm128 leaves + MFMA updates), N^2 n flop"
                                               + ("; timed ALONE in the two-call form after the timed region (in the one-pass step it "
                                                  "runs beside the Cholesky)" if one_pass else "")}
            # the peaks this box sustains on bare probe kernels, and every fraction against them as well
            try:
                pk = probe_peaks(ctx)
                out["peaks_probe"] = pk
                if pk.get("fp64_mfma_tflops", 0) > 0:
                    for key in ("roofline", "solve_v_mfma"):
                        if key in out:
                            out[key]["frac_of_probed"] = out[key]["achieved"] / pk["fp64_mfma_tflops"]
                if "kbuild_hbm" in out and pk.get("hbm_write_gbps", 0) > 0:
                    out["kbuild_hbm"]["frac_of_probed"] = out["kbuild_hbm"]["achieved"] / pk["hbm_write_gbps"]
                if "alpha_hbm" in out and pk.get("hbm_read_gbps", 0) > 0:
                    out["alpha_hbm"]["frac_of_probed"] = out["alpha_hbm"]["achieved"] / pk["hbm_read_gbps"]
            except Exception as e:      # never cost the bench line
                out["peaks_probe"] = {"error": str(e)}
            # the peaks as this box reports them (hipDeviceProp_t), next to the nominal ones the fractions are priced against
            try:
                di = ctx.device_info()
                out["peaks_box"] = {
                    "compute_units": di["compute_units"], "clock_khz": di["clock_khz"], "mem_clock_khz": di["mem_clock_khz"],
                    "mem_bus_bits": di["mem_bus_bits"], "global_mem_gb": di["global_mem_bytes"] / 1e9,
                    "fp64_mfma_tflops": di["compute_units"] * 4 * 32 * di["clock_khz"] * 1e3 / 1e12,
                    "note": "hipDeviceProp_t as read on this box. fp64 matrix peak = CUs x 4 SIMDs x 32 flop/clk (one "
                            "v_mfma_f64_16x16x4 = 2048 flop per 64 cycles) x clock = the nominal 78.6 TF/s. The runtime reports "
                            "the memory bus width and a memory clock but not the HBM3E pin rate, so the HBM peak stays the "
                            "nominal 8 TB/s (8192 bits x 8 Gb/s per pin)"}
            except Exception as e:      # never cost the bench line
                out["peaks_box"] = {"error": str(e)}
        out["targets"] = targets
        if not multi:
            # the wall "including H2D of X, y, Xs and D2H of mu, sigma" -- one step timed around
            # the uploads as well (gpmi_set_train + gpmi_set_test: H2D copies, bounding boxes on the host); never `value`
            wd.beat("step incl. transfers")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.set_train(X, y)
            ctx.set_test(Xs)
            step(args.steps)
            torch.cuda.synchronize()
            t_incl = time.perf_counter() - t1
            out["seconds_incl_transfers"] = t_incl
            out["transfers_note"] = ("one step timed around gpmi_set_train + gpmi_set_test as well (H2D of X, y, X*: %.1f MB; "
                                     "D2H of mu, var, alpha is inside every step): %+.2f ms against ms_per_step"
                                     % ((X.nbytes + y.nbytes + Xs.nbytes) / 1e6, t_incl * 1e3 - ms_per_step))
            if N == 65536 and d == 8 and n == 4096 and not args.no_extra_configs and not force_dist:
                try:
                    out["extra_configs"] = extra_configs(ctx, wd)
                except Exception as e:      # never cost the bench line
                    out["extra_configs"] = {"error": repr(e)}
        wd.beat("cpu baseline")
        wd.stall_s = max(wd.stall_s, 600.0)       # CPU work: no heartbeat inside
        if world == 1 and not args.no_cpu_baseline and not force_dist:
            out["cpu_baseline"] = cpu_baseline(d, n)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
