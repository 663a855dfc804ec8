#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 scripts/cfg5_lanes_la.py 32768 > gpurun_out/r04_cfg5_lanes_la.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_cfg5_lanes_la.txt
timeout -k 10 300 python3 scripts/cfg5_lanes_la.py 24576 > gpurun_out/r04_cfg5_lanes_la_24576.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_cfg5_lanes_la_24576.txt
