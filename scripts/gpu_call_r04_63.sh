#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{ timeout -k 10 200 python3 scripts/one_pass_opts.py 16384 1024; timeout -k 10 300 python3 scripts/one_pass_opts.py 32768 4096; } > gpurun_out/r04_one_pass_opts.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_one_pass_opts.txt
