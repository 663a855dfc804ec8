#!/bin/bash
# next-row rates (f1, f2, f4) + the full GPU suite at the round's final kernels
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 scripts/next_rows_rate.py 4096 16384 32768 65536 > gpurun_out/r04_next_rows_rate.txt 2>&1; echo "rates rc=$?"; cat gpurun_out/r04_next_rows_rate.txt
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_41_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_41_pytest_gpu.txt
