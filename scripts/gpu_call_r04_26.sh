#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for MODE in 0 9; do
  rm -rf gpurun_out/prof_tmp && mkdir -p gpurun_out/prof_tmp
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/prof_tmp/kt -- python3 scripts/trace_one.py 65536 4096 potrf_server=$MODE > gpurun_out/t26_$MODE.out 2> gpurun_out/t26_$MODE.err
  cat gpurun_out/t26_$MODE.out | grep -v amdgpu
  python3 scripts/trail_durations.py $(find gpurun_out/prof_tmp/kt -name "*.db" | head -1) | tee gpurun_out/r04_trail_durations_mode$MODE.txt
done
rm -rf gpurun_out/prof_tmp
