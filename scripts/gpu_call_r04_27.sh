#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 scripts/resident_cost_random.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_resident_cost_random.txt
timeout -k 10 200 python3 scripts/opt_combo.py 65536 4096 "" "potrf_server=9" "potrf_server=25" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_27_server_poll.txt
