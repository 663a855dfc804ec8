#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python3 scripts/replay_warmup_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -E "source look|step  4"; }
run 0 8 5 0 2 0
run 0 8 5 0 0 0
run 7 8 5 0 2 0
timeout -k 10 400 python3 -m pytest tests/test_dist.py tests/test_replay.py -x -q -m gpu 2>&1 | tail -3
