import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y, Xs = O.synthetic_problem(N, 8, 16)
ctx = GPContext(0)
ctx.set_train(X, y)
ctx.set_option("nb", 128)
# a non-PD noise makes the factorisation fail fast in wall time? no: it still runs. Use tiny problem for chol: not possible.
lml = None
try:
    lml = ctx.factorize(1.0, 2.0, 5e-4)
except Exception as e:
    print(e)
print(ctx.timers()["kbuild"], lml)
