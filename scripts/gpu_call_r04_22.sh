#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 scripts/opt_combo.py 65536 4096 "" "potrf_server=1" "potrf_server=9" "potrf_server=3" "potrf_server=5" "potrf_server=17" "potrf_server=33" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_22_server_ablations.txt
