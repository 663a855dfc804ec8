"""In-kernel timeline of the fused panel kernels (gpmi_probe_panel): microseconds per launch and the cycle
stamps of one instrumented launch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
us, st = ctx.probe_panel(0, reps=50, stamps=True)
st = st.astype(np.int64)
print("potrf128: %.1f us per launch; kernel %d cycles begin->end" % (us, st[49] - st[48]))
for j in range(8):
    b = st[6 * j:6 * j + 6]
    nxt = st[6 * (j + 1)] if j < 7 else st[49]
    print("  step %d: factor %5d | ->bar %5d | trsm-mma %5d | ->bar %5d | syrk-mma %5d | to next %5d"
          % (j, b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3], b[5] - b[4], nxt - b[5]))
for m in (128, 512, 4096, 16384, 65536):
    us, st = ctx.probe_panel(1, m=m, reps=30, stamps=True)
    st = st.astype(np.int64)
    print("trsm128 m=%6d: %.1f us per launch; WG0: stage off-diag %d, invert %d, barrier %d, first slab load %d, mma %d, total %d cycles"
          % (m, us, st[57] - st[56], st[58] - st[57], st[59] - st[58], st[60] - st[59], st[61] - st[60], st[62] - st[56]))
