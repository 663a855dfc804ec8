#!/bin/bash
# block width of the Cholesky under the one-pass form, mid sizes
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 200 python3 scripts/nb_sweep_one_pass.py 16384 1024 0 512 768 1024 1536 2048
timeout -k 10 200 python3 scripts/nb_sweep_one_pass.py 24576 4096 0 1024 1536 2048
timeout -k 10 300 python3 scripts/nb_sweep_one_pass.py 32768 4096 0 1024 1536 2048 3072
timeout -k 10 300 python3 scripts/nb_sweep_one_pass.py 49152 4096 0 1024 2048 3072
} > gpurun_out/r04_nb_sweep_one_pass.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_nb_sweep_one_pass.txt
