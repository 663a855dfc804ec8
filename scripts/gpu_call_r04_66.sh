#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "sample_one_pass_errors" > gpurun_out/r04_66_pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r04_66_pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 scripts/sample_one_pass_ab.py > gpurun_out/r04_sample_one_pass_ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_sample_one_pass_ab.txt
