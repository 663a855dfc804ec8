#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 scripts/opt_combo.py 16384 1024 "" "gemm_balance=0" "" "gemm_balance=0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_8_balance_16384.txt
python3 scripts/opt_combo.py 32768 4096 "" "gemm_balance=0" "" "gemm_balance=0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_8_balance_32768.txt
python3 scripts/opt_combo.py 65536 4096 "" "gemm_balance=0" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_8_balance_65536.txt
python3 scripts/opt_combo.py 24576 2048 "" "gemm_balance=0" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_8_balance_24576.txt
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from gaussian_process_amd import GPContext
with GPContext(0) as ctx:
    ctx.set_option("gemm_persist", 0)
    for (M, N, K) in ((13312+128, 13312, 1024), (8192+128, 8192, 1024), (12288+128, 12288, 1024), (4096+128, 4096, 1024), (28672+128, 28672, 2048), (14336+128, 14336, 2048)):
        for bal in (1, 0):
            ctx.set_option("gemm_balance", bal)
            tf, ms = ctx.probe_gemm(M, N, K, 1, 0, 5)
            print("probe %dx%dx%d lower per-tile balance=%d: %.2f TF/s %.3f ms" % (M, N, K, bal, tf, ms), flush=True)
PY
