#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for Q in 8 4; do
  echo "GPU_MAX_HW_QUEUES=$Q"
  GPU_MAX_HW_QUEUES=$Q python3 scripts/opt_combo.py 16384 1024 "" "" 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$Q python3 scripts/opt_combo.py 32768 4096 "" 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$Q python3 scripts/opt_combo.py 65536 4096 "" 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$Q python3 scripts/cfg5.py 2>&1 | grep -v amdgpu.ids
done
