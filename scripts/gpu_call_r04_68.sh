#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_68_pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r04_68_pytest_gpu.txt
