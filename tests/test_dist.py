"""The multi-rank (row-block cyclic) driver: world_size-2 on CPU with gloo and NumPy
stand-ins for the block primitives (runs anywhere), and world_size 2 / 3 on one GPU
with the real HIP primitives (gpu marker; gloo moves the CUDA tensors)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, backend, device, tmp_path, N, d, n, nb, lookahead=1):
    port = _free_port()
    out = str(tmp_path / "res")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", DISTGP_LOOKAHEAD=str(lookahead))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world),
                               port, backend, device, out, str(N), str(d), str(n), str(nb)], env=env)
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [np.load(out + "_rank%d.npz" % r) for r in range(world)]


def _check(res, oracle, N, d, n):
    X, y, Xs = oracle.synthetic_problem(N, d, n, seed=77)
    ell = 2.0 * np.sqrt(d / 8.0)
    ref = oracle.fit_predict_feasible(X, Xs, y, 1.0, ell, 5e-4, use_c=False)
    ref2 = oracle.fit_predict_feasible(X, Xs, y, 1.3, 1.5 * np.sqrt(d / 8.0), 1e-3, use_c=False)
    K = oracle.RBF_kernel(X, X, 1.0, 2.0) - 0.7 * np.eye(N)
    kbad = next(i for i in range(1, N + 1) if np.linalg.eigvalsh(K[:i, :i]).min() <= 0)
    want_batch = [None if s2 < 0 else oracle.fit_predict_feasible(X, Xs[:1], y, sf, l, s2, use_c=False)["lml"]
                  for (l, sf, s2) in res[0]["triples"]]
    # f1 distributed: the posterior-covariance factor and the drop-in prediction(..., dist=gp)
    pref = oracle.posterior(X, Xs, y, 1.0, ell, 5e-4) if N <= 1200 else None
    if pref is not None:
        Lref = np.linalg.cholesky(pref["K_ss"] + 1e-6 * np.eye(n) - pref["v"].T @ pref["v"])
        np.random.seed(123)
        fref = pref["mu"].reshape(-1, 1) + Lref @ np.random.normal(size=(n, 2))
    if N <= 2200:      # f2 distributed against the oracle's statement-by-statement gradient (:43-57, :144)
        _, l_var, sigma_var, alpha_o, K_y = oracle.lml_and_gradient(X, y, 1.0, ell)
        sq = ((X[:, :, None] - X[:, :, None].T) ** 2).sum(1)
        e = np.exp(-.5 * sq / ell ** 2)
        scale_l = .5 * abs(alpha_o @ (e * sq / ell ** 3) @ alpha_o) + .5 * abs(np.sum(K_y * (e * sq / ell ** 3)))
        scale_s = .5 * abs(alpha_o @ (2 * e) @ alpha_o) + .5 * abs(np.sum(K_y * (2 * e)))
        for r in res:
            assert abs(r["g_l"] - l_var) <= 1e-9 * scale_l and abs(r["g_s"] - sigma_var) <= 1e-9 * scale_s
    for r in res:
        if pref is not None:
            assert np.max(np.abs(r["Lp"] - Lref)) <= 1e-6
            assert np.max(np.abs(r["d_fp"] - fref)) <= 1e-6
        assert np.max(np.abs(r["d_mu"] - ref["mu"])) <= 1e-9 and abs(r["d_lml"] - ref["lml"]) <= 1e-10 * abs(ref["lml"])
        assert r["d_cml"] == r["d_lml"]
        assert abs(r["lml"] - ref["lml"]) <= 1e-10 * abs(ref["lml"])
        assert np.max(np.abs(r["mu"] - ref["mu"])) <= 1e-9
        assert r["lml1"] == r["lml"]                      # one pass: the same factor, mean and variance to rounding
        assert np.max(np.abs(r["mu1"] - ref["mu"])) <= 1e-9 and np.max(np.abs(r["var1"] - ref["var"])) <= 1e-10
        assert np.max(np.abs(r["var"] - ref["var"])) <= 1e-10
        assert np.max(np.abs(r["alpha"] - ref["alpha"])) <= 1e-8 * np.max(np.abs(ref["alpha"]))
        assert abs(r["lml2"] - ref2["lml"]) <= 1e-10 * abs(ref2["lml"])
        assert np.max(np.abs(r["mu2"] - ref2["mu"])) <= 1e-9
        assert np.max(np.abs(r["sd2"] - np.sqrt(ref2["var"]))) <= 1e-9
        assert int(r["raised"]) == kbad
        # sharded batch of log-marginal-likelihoods (config 5): per-triple oracle, NaN + status for the bad one
        for t, (l, sf, s2) in enumerate(r["triples"]):
            if s2 < 0:
                assert np.isnan(r["blml"][t]) and r["bst"][t] == 1
            else:
                want = want_batch[t]
                assert abs(r["blml"][t] - want) <= 1e-10 * abs(want) and r["bst"][t] == 0
    for r in res[1:]:                      # every rank returns the same bits
        for key in ("lml", "mu", "var", "lml2", "mu2", "sd2", "blml", "alpha", "Lp", "d_fp", "g_l", "g_s"):
            assert np.array_equal(r[key], res[0][key], equal_nan=True), key


@pytest.mark.parametrize("N,d,n,nb", [(700, 3, 50, 128), (300, 8, 140, 256)])
def test_two_ranks_gloo_cpu(oracle, tmp_path, N, d, n, nb):
    res = _run(2, "gloo", "cpu", tmp_path, N, d, n, nb)
    _check(res, oracle, N, d, n)


def test_three_ranks_gloo_cpu_uneven_blocks(oracle, tmp_path):
    res = _run(3, "gloo", "cpu", tmp_path, 520, 2, 33, 128)      # 5 blocks over 3 ranks
    _check(res, oracle, 520, 2, 33)


@pytest.mark.parametrize("world,N,d,n,nb", [(2, 700, 3, 50, 128), (3, 520, 2, 33, 128), (2, 300, 8, 140, 256),
                                            (3, 1100, 4, 20, 128)])
def test_critical_path_first_schedule_gloo_cpu(oracle, tmp_path, world, N, d, n, nb):
    """lookahead level 2: the diagonal chain on its own stream (same results, different order of launches)."""
    res = _run(world, "gloo", "cpu", tmp_path, N, d, n, nb, lookahead=2)
    _check(res, oracle, N, d, n)


@pytest.mark.parametrize("world,N,d,n,nb,la", [(8, 700, 3, 50, 128, 2),      # 6 blocks over 8 ranks: ranks 6, 7 own none
                                                (8, 1100, 4, 20, 128, 2),     # 9 blocks: one rank with two
                                                (8, 2048, 8, 64, 128, 1),     # 16 blocks, two per rank
                                                (8, 1408, 5, 40, 128, 0),
                                                (4, 1100, 4, 20, 128, 2), (4, 700, 3, 50, 256, 1)])
def test_four_and_eight_ranks_gloo_cpu(oracle, tmp_path, world, N, d, n, nb, la):
    """north_star's own topology (8 ranks; 4 for the scaling curve's middle point) through the same schedule:
    uneven block counts, ranks that own no block at all, the y block on every possible rank."""
    res = _run(world, "gloo", "cpu", tmp_path, N, d, n, nb, lookahead=la)
    _check(res, oracle, N, d, n)


def _thread_ranks(world, device, N, d, n, nb, la, oracle, grad=True):
    """G ranks as G threads of this process (tests/thread_comm.py): fit, predict, alpha, posterior factor, LML gradient,
    a refit on resident data and a not-PD refit, per rank; checked against the oracle and for equal bits on all ranks."""
    import torch
    from thread_comm import ThreadWorld
    from gaussian_process_amd.dist import DistGP
    X, y, Xs = oracle.synthetic_problem(N, d, n, seed=77)
    ell = 2.0 * np.sqrt(d / 8.0)

    def rank_body(r, comm):
        if device == "cpu":
            from numpy_block_ops import NumpyBlockOps
            gp = DistGP(nb=nb, ops=NumpyBlockOps(), lookahead=la, comm=comm)
        else:
            torch.cuda.set_device(0)
            gp = DistGP(0, nb=nb, lookahead=la, comm=comm)
        out = {}
        out["lml"] = gp.fit(X, y, 1.0, ell, 5e-4)
        out["mu"], out["var"] = gp.predict(Xs, want_sd=False)
        out["alpha"] = gp.alpha()
        out["Lp"] = gp.post_chol(1e-6)
        if grad:
            out["g"] = np.array(gp.lml_grad())
        # prediction() in one pass: every rank's share of the test rows rides through the factorisation below its blocks
        out["lml1"], out["mu1"], out["var1"] = gp.fit_predict_resident(1.0, ell, 5e-4, want_sd=False)
        out["alpha1"] = gp.alpha()
        out["Lp1"] = gp.post_chol(1e-6)                   # needs v by columns: runs the two-call sweep first
        out["lml2"] = gp.factorize(1.3, 1.5 * np.sqrt(d / 8.0), 1e-3)
        out["mu2"], out["sd2"] = gp.predict_resident(want_sd=True)
        try:
            gp.factorize(1.0, 2.0, -0.7)
            out["raised"] = 0
        except np.linalg.LinAlgError as e:
            out["raised"] = int(e.bad_pivot)
        return out
    res = ThreadWorld(world).run(rank_body)
    ref = oracle.fit_predict_feasible(X, Xs, y, 1.0, ell, 5e-4, use_c=False)
    ref2 = oracle.fit_predict_feasible(X, Xs, y, 1.3, 1.5 * np.sqrt(d / 8.0), 1e-3, use_c=False)
    K = oracle.RBF_kernel_chunked(X, X, 1.0, 2.0) - 0.7 * np.eye(N)
    try:
        np.linalg.cholesky(K[:256, :256])
        kbad = None
    except np.linalg.LinAlgError:
        kbad = next(i for i in range(1, 257) if np.linalg.eigvalsh(K[:i, :i]).min() <= 0)
    r0 = res[0]
    assert abs(r0["lml"] - ref["lml"]) <= 1e-10 * abs(ref["lml"])
    assert np.max(np.abs(r0["mu"] - ref["mu"])) <= 1e-9 and np.max(np.abs(r0["var"] - ref["var"])) <= 1e-10
    assert np.max(np.abs(r0["alpha"] - ref["alpha"])) <= 1e-8 * np.max(np.abs(ref["alpha"]))
    assert r0["lml1"] == r0["lml"] and np.array_equal(r0["alpha1"], r0["alpha"]) and np.array_equal(r0["Lp1"], r0["Lp"])
    assert np.max(np.abs(r0["mu1"] - ref["mu"])) <= 1e-9 and np.max(np.abs(r0["var1"] - ref["var"])) <= 1e-10
    assert np.max(np.abs(r0["mu1"] - r0["mu"])) <= 1e-11 * max(1.0, 1e-3 * np.max(np.abs(ref["alpha"])))
    assert np.max(np.abs(r0["var1"] - r0["var"])) <= 1e-12
    assert abs(r0["lml2"] - ref2["lml"]) <= 1e-10 * abs(ref2["lml"])
    assert np.max(np.abs(r0["mu2"] - ref2["mu"])) <= 1e-9 and np.max(np.abs(r0["sd2"] - np.sqrt(ref2["var"]))) <= 1e-9
    if kbad is not None:
        assert r0["raised"] == kbad
    else:
        assert r0["raised"] > 0
    if N <= 1200:
        pref = oracle.posterior(X, Xs, y, 1.0, ell, 5e-4)
        Lref = np.linalg.cholesky(pref["K_ss"] + 1e-6 * np.eye(n) - pref["v"].T @ pref["v"])
        assert np.max(np.abs(r0["Lp"] - Lref)) <= 1e-6
    if grad and N <= 2200:
        _, l_var, sigma_var, alpha_o, K_y = oracle.lml_and_gradient(X, y, 1.0, ell)
        sq = ((X[:, :, None] - X[:, :, None].T) ** 2).sum(1)
        e = np.exp(-.5 * sq / ell ** 2)
        scale_l = .5 * abs(alpha_o @ (e * sq / ell ** 3) @ alpha_o) + .5 * abs(np.sum(K_y * (e * sq / ell ** 3)))
        scale_s = .5 * abs(alpha_o @ (2 * e) @ alpha_o) + .5 * abs(np.sum(K_y * (2 * e)))
        assert abs(r0["g"][0] - l_var) <= 1e-9 * scale_l and abs(r0["g"][1] - sigma_var) <= 1e-9 * scale_s
    for r in res[1:]:
        for key in r0:
            assert np.array_equal(r[key], r0[key], equal_nan=True), key
    return res


def test_eight_ranks_as_threads_cpu(oracle):
    """the in-process communicator itself (tests/thread_comm.py), on the NumPy stand-ins: 8 ranks, 6 blocks"""
    _thread_ranks(8, "cpu", 700, 3, 50, 128, 2, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,d,n,nb,la", [(8, 4096, 8, 200, 128, 2),      # 32 blocks, 4 per rank
                                                (8, 4000, 8, 130, 256, 2),      # 16 blocks, ragged N
                                                (8, 4096, 8, 200, 256, 1), (8, 2100, 8, 64, 512, 2),   # 5 blocks: ranks 5-7 own none
                                                (4, 6144, 8, 128, 512, 2), (8, 6144, 16, 128, 256, 0)])
def test_eight_ranks_on_one_gpu_hip(oracle, world, N, d, n, nb, la):
    """north_star's topology with the REAL block primitives: 8 (and 4) ranks as threads of one process sharing the test
    box's one GPU (the pool allows at most six processes on the card, so gloo ranks cannot do this): per-step offset
    tables into an 8-chunk gather buffer, staircase row maps of a rank that owns every eighth block, ranks with no
    block at all, three streams per rank.  Oracle parity and the same bits on every rank."""
    _thread_ranks(world, "cuda", N, d, n, nb, la, oracle, grad=N <= 4096)


def _run_kernels(world, device, tmp_path, nb, lookahead=2):
    port = _free_port()
    out = str(tmp_path / "kres")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", DISTGP_LOOKAHEAD=str(lookahead))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker_kernels.py"), str(r), str(world),
                               port, "gloo", device, out, str(nb)], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [np.load(out + "_rank%d.npz" % r) for r in range(world)]


def _check_kernels(res, oracle):
    """f4 on the partitioned path against the reference's own outputs (kernels_lin_per.npz: prediction(..., 'lin' / 'per')
    of GP_regression.py:125-136; kernels_bo_co2.npz: CO2_example.py's compute_mar_likelihood / make_prediction executed
    from source), the oracle for the square-K_s case, and equal bits on every rank"""
    from conftest import golden
    g, z = golden("kernels_lin_per"), golden("kernels_bo_co2")
    r = res[0]
    assert np.allclose(r["lin_mu"], g["lin_mu"], atol=1e-9) and np.allclose(r["lin_sd"], g["lin_sd"], atol=1e-9)
    assert np.allclose(r["lin_fp"], g["lin_fpost"], atol=1e-6)
    assert np.allclose(r["per_mu"], g["per_mu"], atol=1e-9) and np.allclose(r["per_sd"], g["per_sd"], atol=1e-9)
    assert np.allclose(r["per_fp"], g["per_fpost"], atol=1e-6)
    want = oracle.compute_mar_likelihood(g["X"], None, g["y_lin"], 1.0, 1.3)
    assert abs(r["rbf_lml"] - want) <= 1e-10 * abs(want)
    assert abs(r["co2_lml"] - float(z["co2_lml"])) <= 1e-9 * abs(float(z["co2_lml"]))
    scale = np.abs(z["co2_mu"]).max()
    assert np.allclose(r["co2_mu"], z["co2_mu"], rtol=0, atol=1e-8 * scale)
    assert np.allclose(r["co2_sd"], z["co2_sd"], rtol=0, atol=1e-8, equal_nan=True)
    assert np.allclose(r["co2_fp"], z["co2_fpost"], rtol=0, atol=1e-5 * scale)
    mu, sd, _, _ = oracle.co2_posterior(z["co2_X"], z["co2_X"] + 0.37, z["co2_y"], z["co2_theta"], 5e-4)
    assert np.allclose(r["co2_sq_mu"], mu, rtol=0, atol=1e-8 * scale)
    assert np.allclose(np.sqrt(np.maximum(r["co2_sq_var"], 0)), np.nan_to_num(sd), rtol=0, atol=1e-7)
    assert int(r["grad_refused"]) == 1
    for q in res[1:]:
        for key in r.files:
            assert np.array_equal(q[key], r[key], equal_nan=True), key


@pytest.mark.parametrize("world,nb", [(2, 128), (3, 128), (2, 256)])
def test_other_covariance_functions_on_the_partition_gloo_cpu(oracle, tmp_path, world, nb):
    """prediction(..., 'lin' / 'per', dist=gp) and the CO2 composite kernel through DistGP (SURVEY.md section 8f row f4 on
    the partitioned path), NumPy stand-ins for the block primitives"""
    _check_kernels(_run_kernels(world, "cpu", tmp_path, nb), oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("world,nb,la", [(2, 128, 2), (3, 128, 1), (1, 256, 2)])
def test_other_covariance_functions_on_the_partition_hip(oracle, tmp_path, world, nb, la):
    """the same with the HIP primitives (gpmi_dev_cov_rows / gpmi_dev_cov_cross), ranks sharing the test box's GPU"""
    _check_kernels(_run_kernels(world, "cuda", tmp_path, nb, lookahead=la), oracle)


def test_two_ranks_gloo_cpu_without_lookahead(oracle, tmp_path):
    res = _run(2, "gloo", "cpu", tmp_path, 640, 4, 40, 128, lookahead=0)
    _check(res, oracle, 640, 4, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,d,n,nb,la", [(2, 1500, 8, 200, 256, 1), (3, 2100, 8, 130, 128, 1),
                                                (2, 4096, 8, 512, 512, 1), (2, 1500, 8, 200, 256, 0),
                                                (1, 1300, 8, 100, 256, 1),
                                                (2, 6144, 8, 128, 256, 1),    # large enough for the LDS-DMA GEMM + row map
                                                (1, 1300, 8, 100, 256, 2), (2, 1500, 8, 200, 256, 2),
                                                (3, 2100, 8, 130, 128, 2), (2, 6144, 8, 128, 256, 2),
                                                (2, 6144, 8, 128, 2048, 2), (3, 6144, 8, 128, 1024, 2),   # the bench's block rows
                                                (2, 2304, 16, 150, 256, 2), (3, 2304, 16, 150, 128, 1)])    # cfg4's d
def test_ranks_on_one_gpu_hip(oracle, tmp_path, world, N, d, n, nb, la):
    res = _run(world, "gloo", "cuda", tmp_path, N, d, n, nb, lookahead=la)
    _check(res, oracle, N, d, n)


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs on one node (RCCL between ranks)")
@pytest.mark.parametrize("la", [1, 2])
def test_two_ranks_rccl_one_gpu_each(oracle, tmp_path, la):
    """The real thing where the box has it: two ranks, one GPU each, backend "nccl" (RCCL over xGMI); same checks
    as the gloo runs, every rank the same bits."""
    res = _run(2, "nccl", "cuda_per_rank", tmp_path, 6144, 8, 128, 512, lookahead=la)
    _check(res, oracle, 6144, 8, 128)


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs on one node (RCCL between ranks)")
def test_two_ranks_rccl_through_the_c_abi(oracle, tmp_path):
    """the same with the collectives issued through this library's own RCCL binding (gpmi_comm_*, dist.RcclComm: one
    communicator per stream, no process-group layer); gloo only carries the ncclUniqueIds"""
    os.environ["GPMI_DIST_COMM"] = "rccl"
    try:
        res = _run(2, "gloo", "cuda_per_rank", tmp_path, 6144, 8, 128, 512, lookahead=2)
    finally:
        os.environ.pop("GPMI_DIST_COMM", None)
    _check(res, oracle, 6144, 8, 128)


@pytest.mark.gpu
def test_rccl_through_the_c_abi_on_a_world_of_one():
    """gpmi_comm_* (librccl opened at run time by libgpmi355x.so, collectives on the caller's streams, one communicator
    per stream): every collective of the multi-rank schedule on communicators of size 1 on the one GPU of the test box;
    results equal the run without collectives bit for bit, for every lookahead level"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_world1.py"), "3072", "rccl"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "RCCL world-1 path: OK" in p.stdout and "rccl through the C-ABI" in p.stdout


@pytest.mark.gpu
def test_rccl_collectives_on_a_world_of_one():
    """The RCCL call path itself (backend "nccl": broadcast, all_gather_into_tensor, int64 MIN all_reduce,
    asynchronous stream semantics over the three streams of the schedule) on the one GPU of the test box:
    DistGP(force_collectives=True) issues every collective of the multi-rank schedule on a communicator
    of size 1; the results must equal the run without collectives bit for bit, for every lookahead level."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=_free_port())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_world1.py"), "3072"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "RCCL world-1 path: OK" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,band_rows", [(2048, 4096, 256, 128), (3072, 6144, 512, 256), (512, 1024, 128, 128)])
def test_row_map_update_live_supertiles_only(M, N, K, band_rows):
    """gpmi_dev_gemm_nt_rowmap_host (only the supertiles with live tiles are launched) against the plain
    row-map launch (whole rectangle) and NumPy, on staircases like a rank's stacked row blocks: bitwise
    equal to each other, columns beyond a band's reach untouched."""
    import torch
    from gaussian_process_amd.dist import HipBlockOps
    ops = HipBlockOps(0)
    rng = np.random.default_rng(M + K)
    nb = M // band_rows
    reach = np.minimum(N, 384 + 2 * band_rows * np.arange(nb)).astype(np.int32)     # slope 2, like 2 ranks
    reach[-1] = N                                                                  # the y rows reach everything
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K))
    C0 = rng.standard_normal((M, N))
    want = C0.copy()
    for b in range(nb):
        rows = slice(b * band_rows, (b + 1) * band_rows)
        live = -(-int(reach[b]) // 128) * 128                                      # whole 128-column tiles
        live = min(live, N)
        want[rows, :live] -= A[rows] @ B[:live].T
    dev = torch.device("cuda", 0)
    Ad, Bd = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    rm = torch.from_numpy(reach).to(dev)
    outs = []
    for host in (None, reach):
        Cd = torch.from_numpy(C0).to(dev)
        ops.gemm_nt_rowmap(Cd, Ad, Bd, rm, band_rows, host)
        torch.cuda.synchronize()
        outs.append(Cd.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.allclose(outs[1], want, rtol=0, atol=1e-10 * np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,brows,with_map", [(1024, 2048, 256, 256, True), (2560, 3072, 512, 512, True),
                                                  (512, 1024, 128, 128, False), (384, 768, 1024, 128, True)])
def test_block_table_update_reads_the_gather_buffer_in_natural_order(M, N, K, brows, with_map):
    """gpmi_dev_gemm_nt_blocks: B scattered as row blocks in a flat buffer (the all-gather's receive buffer: one chunk
    per rank) read through an offset table == the same update on the assembled B, bit for bit."""
    import torch
    from gaussian_process_amd.dist import HipBlockOps
    ops = HipBlockOps(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(M + N + K)
    nblk = N // brows
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K))
    C0 = rng.standard_normal((M, N))
    perm = rng.permutation(nblk + 3)[:nblk]                      # where each natural block sits in the flat buffer
    flat = np.full(((nblk + 3) * brows, K), np.nan)
    for i, q in enumerate(perm):
        flat[q * brows:(q + 1) * brows] = B[i * brows:(i + 1) * brows]
    boff = torch.from_numpy((perm * brows * K).astype(np.int64)).to(dev)
    reach_h = rm = None
    if with_map:
        bands = M // 128
        reach_h = np.minimum(N, 256 + 3 * 128 * np.arange(bands)).astype(np.int32)
        reach_h[-1] = N
        rm = torch.from_numpy(reach_h).to(dev)
    Ad, Bd, Fd = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(flat.reshape(-1)).to(dev)
    C1 = torch.from_numpy(C0).to(dev)
    ops.gemm_nt_blocks(C1, Ad, Fd, K, boff, brows, rm, 128, reach_h)
    C2 = torch.from_numpy(C0).to(dev)
    if with_map:
        ops.gemm_nt_rowmap(C2, Ad, Bd, rm, 128, reach_h)
    else:
        ops.gemm_nt(C2, Ad, Bd)
    torch.cuda.synchronize()
    got, ref = C1.cpu().numpy(), C2.cpu().numpy()
    assert np.all(np.isfinite(got))
    if M // 128 * (N // 128) >= 256:
        assert np.array_equal(got, ref)                           # same kernel, same tiles: identical bits
    else:
        assert np.allclose(got, ref, rtol=0, atol=1e-11 * np.abs(ref).max())   # the plain launch takes the small-tile kernel


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return p


def test_bench_spawns_its_own_ranks_cpu():
    """`python bench.py --gpus N` outside torch.distributed.run starts N fresh rank processes itself (the driver's
    invocation).  GPMI_BENCH_SPAWN_ECHO makes every rank report its environment and exit before it imports torch."""
    import json
    p = _bench(["--gpus", "3"], {"GPMI_BENCH_SPAWN_ECHO": "1"}, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    seen = sorted((json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")), key=lambda r: r["rank"])
    assert [r["rank"] for r in seen] == [0, 1, 2] and all(r["world"] == 3 and r["local_rank"] == r["rank"] for r in seen)
    assert len({r["port"] for r in seen}) == 1 and all(r["addr"] == "127.0.0.1" for r in seen)
    # a failing rank makes the parent exit non-zero
    p = _bench(["--gpus", "2"], {"GPMI_BENCH_SPAWN_ECHO": "fail1"}, timeout=120)
    assert p.returncode != 0


@pytest.mark.gpu
def test_bench_two_ranks_through_the_self_spawn_path():
    """The driver's multi-GPU command line, rehearsed on the one GPU of the test box: gloo moves the panels, both
    ranks share device 0.  One JSON line from rank 0 with n_gpus = rccl_ranks = 2."""
    import json
    p = _bench(["--gpus", "2", "--size", "8192", "--ntest", "512", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
               {"GPMI_BENCH_BACKEND": "gloo", "GPMI_BENCH_ONE_DEVICE": "1"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["partition"] == "row-block cyclic x2"
    assert out["value"] > 0 and np.isfinite(out["lml"]) and len(out["per_rank_ms"]) == 2
    # the same problem on the single-GPU path: the same log-marginal-likelihood to the path's LML tolerance (the two
    # factorisations block differently -- 512 rows per block here, 512-column panels there -- and the LML is a sum of
    # cancelling terms a hundred times its size: 4e-12 relative measured)
    p1 = _bench(["--gpus", "1", "--size", "8192", "--ntest", "512", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {})
    assert p1.returncode == 0, p1.stderr[-3000:]
    one = json.loads([l for l in p1.stdout.splitlines() if l.startswith("{")][0])
    assert abs(one["lml"] - out["lml"]) <= 1e-10 * abs(one["lml"])
    # the bench contract's keys on the single-GPU line (roofline blocks of the trailing update, a1+a2, a5, a7)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "targets", "kbuild_hbm", "alpha_hbm", "solve_v_mfma",
                "stages_ms"):
        assert key in one, key
    assert one["dtype"] == "f64" and one["config"]["workload"] and one["roofline"]["bound"] == "mfma"
    assert 0 < one["alpha_hbm"]["frac"] < 1 and 0 < one["solve_v_mfma"]["frac"] < 1
    # the step is prediction() in one pass; the two-call form is timed beside it and gives the same bits
    assert one["config"]["call_form"].startswith("one pass") and out["config"]["call_form"].startswith("one pass")
    oc = one["other_call_form"]
    assert oc["form"].startswith("two calls") and oc["ms_per_step"] > 0
    assert oc["lml_equal"] and oc["alpha_equal"] and oc["max_abs_dmu"] == 0.0 and oc["max_abs_dvar"] == 0.0


@pytest.mark.gpu
def test_bench_under_torch_distributed_run():
    """the driver's other launch form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (each process
    is one rank; no self-spawn), rehearsed on one GPU with gloo"""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPMI_BENCH_BACKEND="gloo", GPMI_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--size", "4096", "--ntest", "256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["value"] > 0


def test_multi_rank_bench_cannot_hang_silently(tmp_path):
    """bench.py's watchdog: a rank that stalls in front of a collective must end the run with a
    non-zero code inside the deadline and with the stalled rank, step and phase on stderr.  The bench's skeleton on
    gloo / CPU tensors (GPMI_BENCH_REHEARSE=gloo_cpu), rank 1 sleeping in step 1's alpha phase."""
    import time
    env = dict(os.environ, GPMI_BENCH_REHEARSE="gloo_cpu", GPMI_BENCH_STALL_RANK="1", GPMI_BENCH_STALL_STEP="1",
               GPMI_BENCH_STALL_S="6", GPMI_BENCH_PG_TIMEOUT_S="12", GPMI_BENCH_DEADLINE_S="60", GPMI_BENCH_GRACE_S="8",
               OMP_NUM_THREADS="1")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    dt = time.monotonic() - t0
    assert p.returncode != 0 and dt < 60, (p.returncode, dt)
    assert "[bench watchdog] rank 1: no progress" in p.stderr and "since 'step 1 alpha'" in p.stderr, p.stderr[-2000:]
    beats1 = [ln for ln in p.stderr.splitlines() if ln.startswith("[bench] rank 1 ")]
    assert beats1[-1] == "[bench] rank 1 step 1 alpha"
    assert "[bench parent] rank" in p.stderr
    assert p.stdout.strip() == ""                        # no result line from a failed run
    # the same run without a stalled rank ends cleanly with its one line
    env.pop("GPMI_BENCH_STALL_RANK")
    ok = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                        env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and '"rehearsal": "gloo_cpu"' in ok.stdout, ok.stderr[-2000:]


def test_multi_rank_bench_parent_deadline(tmp_path):
    """every rank wedged (no watchdog can help a process that never gets to run it: stall limit set far above the
    deadline): the parent kills them at GPMI_BENCH_DEADLINE_S and exits 124"""
    import time
    env = dict(os.environ, GPMI_BENCH_REHEARSE="gloo_cpu", GPMI_BENCH_STALL_RANK="0", GPMI_BENCH_STALL_STEP="0",
               GPMI_BENCH_STALL_S="600", GPMI_BENCH_PG_TIMEOUT_S="600", GPMI_BENCH_DEADLINE_S="10", OMP_NUM_THREADS="1")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 124 and time.monotonic() - t0 < 40, (p.returncode, p.stderr[-1500:])
    assert "[bench parent] deadline of 10 s reached" in p.stderr
