"""Bare HBM store patterns (gpmi_probe_hbm_ex): ceiling for the K build's write stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
gb = 32
T = 65536 // 128
for mode in (6, 5, 2):
    for blocks in (1280, 16384, 65536):
        v = ctx.probe_hbm_ex(gb << 30, mode, blocks)
        if mode == 6:
            v *= 0.5 * (1 + 1.0 / T)       # only the lower tiles are written
        print("buffer %d GiB mode %d blocks %6d: %.0f GB/s" % (gb, mode, blocks, v), flush=True)
