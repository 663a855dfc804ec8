#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 scripts/opt_combo.py 65536 4096 "" "potrf_server=73" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_28_a.txt
timeout -k 10 200 python3 scripts/opt_combo.py 65536 4096 "potrf_server=137" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_28_b.txt
timeout -k 10 200 python3 scripts/opt_combo.py 65536 4096 "potrf_server=201" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_28_c.txt
