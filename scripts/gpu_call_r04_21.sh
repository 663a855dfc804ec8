#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 150 python3 scripts/opt_combo.py 16384 1024 "" "potrf_server=1" "" "potrf_server=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_21_server_16384.txt || exit 1
timeout -k 10 150 python3 scripts/opt_combo.py 32768 4096 "" "potrf_server=1" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_21_server_32768.txt || exit 1
timeout -k 10 150 python3 scripts/opt_combo.py 65536 4096 "" "potrf_server=1" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_21_server_65536.txt
