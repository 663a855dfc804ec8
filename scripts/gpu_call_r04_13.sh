#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python3 scripts/replay_warmup_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -E "source look|step  4"; }
run 0 8 5 0 2 0
run 0 8 5 0 0 0
run 0 8 5 0 0 1
run 0 8 5 0 0 2
run 0 8 5 0 0 3
run 0 8 5 0 0 4
run 0 8 5 0 2 2
run 0 8 5 0 2 4
GPU_MAX_HW_QUEUES=8 run 0 8 5 0 2 0
GPU_MAX_HW_QUEUES=8 run 0 8 5 0 0 0
