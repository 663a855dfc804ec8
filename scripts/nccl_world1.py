"""Every collective of the multi-rank schedule on a real RCCL communicator of size 1 (one GPU):
DistGP(force_collectives=True) issues the broadcasts / all-gathers / all-reduces it would issue with
more ranks; results must equal the same run without collectives bit for bit.
  python scripts/nccl_world1.py N [torch|rccl]     torch: RCCL through torch.distributed's "nccl" process group;
                                                   rccl: RCCL through this library's own C-ABI (gpmi_comm_*, dist.RcclComm)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("MASTER_PORT", "29621")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
import gp_oracle as O
from gaussian_process_amd.dist import DistGP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
which = sys.argv[2] if len(sys.argv) > 2 else "torch"
torch.cuda.set_device(0)
if which == "torch":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    make_comm = lambda: None
else:
    from gaussian_process_amd.dist import RcclComm
    make_comm = lambda: RcclComm(0)            # a world of one rank: no side channel needed for the id
    print("comm:", RcclComm(0).describe(), flush=True)
X, y, Xs = O.synthetic_problem(N, 8, 300)
ok = True
for la in (2, 1, 0):
    res = []
    for force in (False, True):
        gp = DistGP(0, nb=256, lookahead=la, force_collectives=force, comm=make_comm() if force or which != "torch" else None)
        lml = gp.fit(X, y, 1.0, 2.0, 5e-4)
        mu, var = gp.predict(Xs, want_sd=False)
        alpha = gp.alpha()
        try:
            gp.factorize(1.0, 2.0, -0.7); bad = 0
        except np.linalg.LinAlgError as e:
            bad = e.bad_pivot
        res.append((lml, mu, var, alpha, bad))
        del gp
    same = (res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
            and np.array_equal(res[0][3], res[1][3]) and res[0][4] == res[1][4])
    ok &= same
    print("lookahead=%d: lml %.9f  bad pivot %d  forced collectives == none: %s" % (la, res[1][0], res[1][4], same), flush=True)
if which == "torch":
    dist.barrier()
    dist.destroy_process_group()
print("RCCL world-1 path:", "OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
