#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from gaussian_process_amd import GPContext
with GPContext(0) as ctx:
    ctx.set_option("gemm_persist", 0)
    for (M, N, K) in ((32768+128, 32768, 2048), (28672+128, 28672, 2048), (14336+128, 14336, 2048), (16384+128, 16384, 1024), (13312+128, 13312, 1024), (8192+128, 8192, 1024), (4096+128, 4096, 1024)):
        for bal in (1, 0):
            ctx.set_option("gemm_balance", bal)
            tf, ms = ctx.probe_gemm(M, N, K, 1, 32, 5)
            print("probe %dx%dx%d lower per-tile balance=%d: %.2f TF/s %.3f ms" % (M, N, K, bal, tf, ms), flush=True)
PY
python3 scripts/opt_combo.py 16384 1024 "" "" 2>&1 | grep -v amdgpu.ids
python3 scripts/opt_combo.py 32768 4096 "" "" 2>&1 | grep -v amdgpu.ids
python3 scripts/opt_combo.py 65536 4096 "" "" 2>&1 | grep -v amdgpu.ids
python3 scripts/cfg5.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or cfg1 or ticket or persistent or properties" 2>&1 | tail -3
