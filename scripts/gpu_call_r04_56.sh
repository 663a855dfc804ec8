#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 scripts/la_min_batch.py > gpurun_out/r04_la_min_batch.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_la_min_batch.txt
