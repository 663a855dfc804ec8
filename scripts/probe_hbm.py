import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for gb in (16,):
    for mode in (5, 2, 0, 3):
        for blocks in (1280, 2048, 16384, 131072):
            print("bytes %d GiB mode %d blocks %6d: %.0f GB/s (nominal bytes)" % (gb, mode, blocks, ctx.probe_hbm_ex(gb << 30, mode, blocks)), flush=True)
