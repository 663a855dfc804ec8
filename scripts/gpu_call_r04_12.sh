#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for PAD in 0 2 130 1058 4100 40000; do
  python3 scripts/replay_warmup_probe.py 0 8 5 $PAD 2>&1 | grep -v amdgpu.ids | grep -E "pad|step  4"
done
