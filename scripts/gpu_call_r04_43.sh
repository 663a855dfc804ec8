#!/bin/bash
# prediction() in one pass: the new tests, then the A/B against the two-call form
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "one_pass" > gpurun_out/r04_43_pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r04_43_pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 scripts/one_pass_ab.py > gpurun_out/r04_one_pass_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r04_one_pass_ab.txt
