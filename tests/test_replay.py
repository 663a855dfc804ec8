"""One rank of a G-rank DistGP run alone (gaussian_process_amd/replay.py: bench.py --replay-rank r --of G).

The replayed rank runs the UNCHANGED schedule of dist.DistGP; its collectives are served out of a stored
factorisation of the same problem.  Checked here: for every rank r of a world of G the replay reproduces that
rank's rows of L, m, the full alpha / mean / variance / LML of the source factorisation, and those agree with the
oracle -- on the NumPy stand-ins (CPU) and, marked gpu, with the HIP block primitives."""
import numpy as np
import pytest


def _replay_all_ranks(oracle, device, G, N, d, n, nb, la, layout="snake"):
    from gaussian_process_amd.dist import block_layout
    from gaussian_process_amd.replay import ReplaySource, replay_rank, check_rank
    ops_of = None
    if device == "cpu":
        from numpy_block_ops import NumpyBlockOps
        ops_of = NumpyBlockOps
    X, y, Xs = oracle.synthetic_problem(N, d, n, seed=77)
    ell = 2.0 * np.sqrt(d / 8.0)
    src = ReplaySource(0, nb, X, y, Xs, 1.0, ell, 5e-4, ops=ops_of() if ops_of else None, lookahead=la)
    ref = oracle.fit_predict_feasible(X, Xs, y, 1.0, ell, 5e-4, use_c=False)
    assert abs(src.lml - ref["lml"]) <= 1e-10 * abs(ref["lml"])
    assert np.max(np.abs(src.mu - ref["mu"])) <= 1e-9 and np.max(np.abs(src.var - ref["var"])) <= 1e-10
    amax = np.max(np.abs(ref["alpha"]))
    for r in range(G):
        gp = replay_rank(0, src, r, G, X, y, Xs, lookahead=la, ops=ops_of() if ops_of else None, layout=layout)
        for rep in range(2):                                  # a second step on resident data gives the same answers
            lml = gp.factorize(1.0, ell, 5e-4)
            alpha = gp.alpha()
            mu, var = gp.predict_resident(want_sd=False)
            chk = check_rank(gp, src)
            assert chk["L_rel"] <= 1e-12, (r, chk)
            if "m_rel" in chk:
                assert chk["m_rel"] <= 1e-11, (r, chk)
            assert abs(lml - src.lml) <= 1e-12 * abs(src.lml), r
            assert abs(lml - ref["lml"]) <= 1e-10 * abs(ref["lml"]), r
            assert np.max(np.abs(mu - ref["mu"])) <= 1e-9 and np.max(np.abs(var - ref["var"])) <= 1e-10, r
            assert np.max(np.abs(alpha - ref["alpha"])) <= 1e-8 * amax, r
            assert np.max(np.abs(alpha - src.alpha_h)) <= 1e-10 * amax, r
        # the one-pass form: this rank's test rows ride below its blocks (the same L; its own points' mean and variance are
        # its own work, the other ranks' come from the source)
        lml1, mu1, var1 = gp.fit_predict_resident(1.0, ell, 5e-4, want_sd=False)
        assert lml1 == lml, r
        assert check_rank(gp, src)["L_rel"] <= 1e-12, r
        assert np.max(np.abs(mu1 - ref["mu"])) <= 1e-9 and np.max(np.abs(var1 - ref["var"])) <= 1e-10, r
        # the absent ranks' messages had the sizes a real rank receives: per fit, the panel column below every block
        # column minus this rank's own share
        T = src.T
        own = block_layout(T, G, layout)[0]
        want = sum(1 for k in range(T - 1) for b in range(k + 1, T) if own[b] != r)
        assert gp.comm.bytes["allgather"] == 3 * want * nb * nb * 8, r          # three fits: the riding rows add no message


@pytest.mark.parametrize("G,N,d,n,nb,la,layout", [(2, 700, 3, 50, 128, 2, "snake"), (3, 520, 2, 33, 128, 2, "snake"),
                                                  (8, 1100, 4, 20, 128, 2, "snake"), (8, 700, 3, 50, 128, 1, "cyclic"),
                                                  (4, 1100, 4, 20, 256, 0, "cyclic"), (4, 2300, 4, 20, 128, 2, "snake"),
                                                  (8, 1100, 4, 20, 128, 2, "balanced"), (3, 1700, 2, 33, 128, 1, "balanced")])
def test_replay_every_rank_cpu(oracle, G, N, d, n, nb, la, layout):
    _replay_all_ranks(oracle, "cpu", G, N, d, n, nb, la, layout)


def test_snake_layout_balances_the_update():
    """block_layout: every block owned once, local indices consistent, and the snake's worst rank within 3 % of the mean
    share of the update at north_star's shape (64 blocks, 8 ranks) where cyclic dealing is 17 % over"""
    from gaussian_process_amd.dist import block_layout
    for T, G in ((64, 8), (32, 4), (6, 8), (9, 8), (5, 3), (1, 2), (128, 8)):
        for layout in ("cyclic", "snake", "balanced"):
            own, li, blocks = block_layout(T, G, layout)
            assert sorted(b for bl in blocks for b in bl) == list(range(T)) and len(own) == T + 1
            for r in range(G):
                assert blocks[r] == sorted(blocks[r]) and all(own[b] == r and li[b] == j for j, b in enumerate(blocks[r]))
                assert abs(len(blocks[r]) - T / G) < 1
            for g0 in range(0, T + 1, G):          # one block per rank in every group of G: row counts differ by <= 1 block
                grp = [own[b] for b in range(g0, min(T + 1, g0 + G))]
                assert len(set(grp)) == len(grp)
            assert li[T] == len(blocks[own[T]])
    share = lambda bl: sum(b * (b + 1) // 2 for b in bl)
    for layout, worst in (("cyclic", 1.169), ("snake", 1.026), ("balanced", 1.004)):
        _, _, blocks = block_layout(64, 8, layout)
        w = [share(bl) for bl in blocks]
        assert abs(max(w) / (sum(w) / 8) - worst) < 2e-3, (layout, max(w) / (sum(w) / 8))


@pytest.mark.gpu
@pytest.mark.parametrize("G,N,d,n,nb,la", [(8, 4096, 8, 200, 128, 2), (8, 4000, 8, 130, 256, 2), (2, 6144, 8, 128, 512, 2),
                                           (4, 6144, 16, 128, 256, 1), (8, 2100, 8, 64, 512, 2)])
def test_replay_every_rank_hip(oracle, G, N, d, n, nb, la):
    """the same with the HIP block primitives on the one GPU of the box"""
    _replay_all_ranks(oracle, "cuda", G, N, d, n, nb, la)


def test_replay_refuses_what_it_cannot_serve(oracle):
    from numpy_block_ops import NumpyBlockOps
    from gaussian_process_amd.replay import ReplaySource, replay_rank
    X, y, Xs = oracle.synthetic_problem(300, 2, 10, seed=3)
    src = ReplaySource(0, 128, X, y, Xs, 1.0, 1.0, 5e-4, ops=NumpyBlockOps())
    gp = replay_rank(0, src, 1, 2, X, y, Xs, ops=NumpyBlockOps())
    gp.factorize(1.0, 1.0, 5e-4)
    gp.predict_resident()
    with pytest.raises(NotImplementedError):
        gp.post_chol(1e-6)                                    # an all-reduce of v^T v: not part of the replayed step
    with pytest.raises(ValueError):
        from gaussian_process_amd.replay import ReplayComm
        ReplayComm(2, 2, src)
