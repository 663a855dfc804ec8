#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "sample or dropin or golden or posterior or post_chol" > gpurun_out/r04_69_pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r04_69_pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 scripts/prediction_rate.py > gpurun_out/r04_prediction_rate.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_prediction_rate.txt
