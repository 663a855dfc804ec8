"""BASELINE config 5 on one GPU: 64 (l, sigma_f, sigma_n^2) triples at N=32768, d=8 through gpmi_lml_batch
(on 8 GPUs dist.sharded_lml_batch gives each rank 8 of them, no data-path collective)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y, _ = O.synthetic_problem(N, 8, 4)
triples = np.array([[l, sf, s2] for l in (1., 2., 3., 4.) for sf in (.5, 1., 1.5, 2.) for s2 in (1e-4, 5e-4, 1e-3, 5e-3)])
ctx = GPContext(0)
ctx.set_train(X, y)
ctx.lml_batch(triples[:2])
t0 = time.perf_counter(); lml, st = ctx.lml_batch(triples); dt = time.perf_counter() - t0
one = ctx.fit(X, y, 1.0, 2.0, 5e-4)
i = [k for k, t in enumerate(triples) if tuple(t) == (2.0, 1.0, 5e-4)][0]
print("N=%d: %d triples in %.2f s (%.3f s each, %.1f TFLOP/s on N^3/3 each); failed %d; entry (2,1,5e-4) == single call: %s"
      % (N, len(triples), dt, dt / len(triples), len(triples) * N ** 3 / 3 / dt / 1e12, int(st.sum()), lml[i] == one), flush=True)
