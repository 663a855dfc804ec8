"""One rank of the multi-rank DistGP tests.  argv: rank world port backend device out_prefix N d n nb"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    backend, device, out = sys.argv[4], sys.argv[5], sys.argv[6]
    N, d, n, nb = (int(v) for v in sys.argv[7:11])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    # a wedged collective must not hang the suite: stacks of every thread and exit after the deadline, and a finite
    # process-group timeout (the same watchdog ideas as bench.py)
    import datetime
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("DISTGP_WORKER_DEADLINE_S", "540")), exit=True)
    import torch
    import torch.distributed as dist
    import gp_oracle as O
    from gaussian_process_amd.dist import DistGP
    pg_timeout = datetime.timedelta(seconds=float(os.environ.get("DISTGP_WORKER_PG_TIMEOUT_S", "300")))
    if backend == "nccl":
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", rank), timeout=pg_timeout)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
    if device == "cpu":
        torch.set_num_threads(2)
        from numpy_block_ops import NumpyBlockOps
        gp = DistGP(nb=nb, ops=NumpyBlockOps(), lookahead=int(os.environ.get("DISTGP_LOOKAHEAD", "1")))
    elif device == "cuda_per_rank":        # one GPU per rank (boxes with >= world GPUs): the real RCCL path
        torch.cuda.set_device(rank)
        gp = DistGP(rank, nb=nb, lookahead=int(os.environ.get("DISTGP_LOOKAHEAD", "1")))
    else:
        torch.cuda.set_device(0)           # all ranks share the one GPU of the test box
        gp = DistGP(0, nb=nb, lookahead=int(os.environ.get("DISTGP_LOOKAHEAD", "1")))
    X, y, Xs = O.synthetic_problem(N, d, n, seed=77)
    lml = gp.fit(X, y, 1.0, 2.0 * np.sqrt(d / 8.0), 5e-4)
    mu, var = gp.predict(Xs, want_sd=False)
    alpha = gp.alpha()
    Lp = gp.post_chol(1e-6)                                       # f1 distributed (GP_regression.py:153-154)
    g_l, g_s = gp.lml_grad()                                      # f2 distributed (tune_hyperparms_regression.py:43-57)
    # prediction() in one pass: each rank's share of the test rows rides through the factorisation (no message of its own)
    lml1, mu1, var1 = gp.fit_predict_resident(1.0, 2.0 * np.sqrt(d / 8.0), 5e-4, want_sd=False)
    # the drop-in surface routed to the multi-rank driver (additive dist= / n_gpus= keywords)
    from gaussian_process_amd import GP_regression as G
    from gaussian_process_amd import tune_hyperparms_regression as T
    np.random.seed(123)
    d_mu, d_sd, d_fp, d_lml = G.prediction(X, Xs, y, 'rbf', 2.0 * np.sqrt(d / 8.0), 2, return_lml=True, dist=gp)
    d_cml = T.compute_mar_likelihood(X, None, y, 1, 2.0 * np.sqrt(d / 8.0), dist=gp)
    gp.set_test(Xs)
    lml2 = gp.factorize(1.3, 1.5 * np.sqrt(d / 8.0), 1e-3)        # refit on resident data
    mu2, sd2 = gp.predict_resident(want_sd=True)
    raised = 0
    try:
        gp.factorize(1.0, 2.0, -0.7)
    except np.linalg.LinAlgError as e:
        raised = int(e.bad_pivot)
    # config-5 style batch: triples sharded over ranks, no data-path collective
    from gaussian_process_amd.dist import sharded_lml_batch
    triples = np.array([[l, sf, s2] for l in (1.0, 2.0) for sf in (0.7, 1.0, 1.4) for s2 in (5e-4,)] + [[2.0, 1.0, -0.7]])
    if device == "cpu":
        def evaluate(tr):
            vals, st = [], []
            for l, sf, s2 in tr:
                try:
                    vals.append(O.compute_mar_likelihood(X, None, y, sf, l, s=s2)); st.append(0)
                except np.linalg.LinAlgError:
                    vals.append(np.nan); st.append(1)
            return np.array(vals), np.array(st)
    else:
        from gaussian_process_amd import GPContext
        ctx = GPContext(rank if device == "cuda_per_rank" else 0)
        ctx.set_train(X, y)
        evaluate = ctx.lml_batch
    blml, bst = sharded_lml_batch(triples, evaluate)
    np.savez(out + "_rank%d.npz" % rank, lml=lml, mu=mu, var=var, lml2=lml2, mu2=mu2, sd2=sd2, raised=raised,
             blml=blml, bst=bst, triples=triples, alpha=alpha, Lp=Lp, d_mu=d_mu, d_sd=d_sd, d_fp=d_fp, d_lml=d_lml,
             d_cml=d_cml, g_l=g_l, g_s=g_s, lml1=lml1, mu1=mu1, var1=var1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
