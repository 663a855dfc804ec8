#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_61_pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r04_61_pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 scripts/small_sizes.py > gpurun_out/r04_small_sizes.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_small_sizes.txt | tail -20
