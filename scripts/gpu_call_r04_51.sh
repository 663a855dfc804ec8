#!/bin/bash
# the whole GPU suite, then the round's evidence at 2b5e842
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_51_pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r04_51_pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
GPMI_GIT_REV=2b5e842 bash scripts/collect_r04.sh r04c
