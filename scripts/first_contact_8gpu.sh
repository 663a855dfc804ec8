#!/bin/bash
# First contact with real RCCL on a multi-GPU node: smallest thing first, stop at the first failure.  Every stage is
# bounded (bench.py's own deadline / watchdog, plus `timeout -k`), prints where it is, and leaves its output under
# gpurun_out/first_contact/.  Run from the repository root on a box with >= 2 (ideally 8) GPUs:
#   bash scripts/first_contact_8gpu.sh [NGPUS]
set -o pipefail
NG=${1:-8}
OUT=gpurun_out/first_contact
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPMI_BENCH_DEADLINE_S=${GPMI_BENCH_DEADLINE_S:-300} GPMI_BENCH_STALL_S=${GPMI_BENCH_STALL_S:-90}
say() { echo "[first contact] $*" | tee -a $OUT/log.txt; }

say "1/4 two ranks, one GPU each, RCCL: tests/test_dist.py::test_two_ranks_rccl_one_gpu_each (+ the C-ABI binding)"
timeout -k 10 600 python3 -m pytest tests/test_dist.py -x -q -m gpu -k "two_ranks_rccl" > $OUT/1_pytest.txt 2>&1 \
    || { say "FAILED at stage 1 (see $OUT/1_pytest.txt)"; tail -30 $OUT/1_pytest.txt; exit 1; }
tail -3 $OUT/1_pytest.txt

say "2/4 bench.py --gpus 2 at N=16384 (one step)"
timeout -k 10 400 python3 bench.py --gpus 2 --size 16384 --ntest 1024 --steps 1 --warmup 0 > $OUT/2_bench_g2.json 2> $OUT/2_bench_g2.err \
    || { say "FAILED at stage 2 (see $OUT/2_bench_g2.err: the last '[bench] rank r step k phase' lines name the collective)"; tail -30 $OUT/2_bench_g2.err; exit 2; }
cut -c1-300 $OUT/2_bench_g2.json

say "3/4 bench.py --gpus $NG at N=16384 (one step)"
timeout -k 10 400 python3 bench.py --gpus $NG --size 16384 --ntest 1024 --steps 1 --warmup 0 > $OUT/3_bench_gN_small.json 2> $OUT/3_bench_gN_small.err \
    || { say "FAILED at stage 3 (see $OUT/3_bench_gN_small.err)"; tail -40 $OUT/3_bench_gN_small.err; exit 3; }
cut -c1-300 $OUT/3_bench_gN_small.json

say "3b/4 the same through this library's own RCCL binding (GPMI_DIST_COMM=rccl; not fatal: the line above is the default path)"
GPMI_DIST_COMM=rccl timeout -k 10 400 python3 bench.py --gpus $NG --size 16384 --ntest 1024 --steps 1 --warmup 0 > $OUT/3b_bench_gN_small_cabi.json 2> $OUT/3b_bench_gN_small_cabi.err \
    && cut -c1-300 $OUT/3b_bench_gN_small_cabi.json || { say "stage 3b failed (see $OUT/3b_bench_gN_small_cabi.err); continuing with the default backend"; tail -20 $OUT/3b_bench_gN_small_cabi.err; }

say "4/4 the real line: bench.py --gpus $NG (N=65536)"
timeout -k 10 560 python3 bench.py --gpus $NG --steps 3 --warmup 1 > $OUT/4_bench_gN.json 2> $OUT/4_bench_gN.err \
    || { say "FAILED at stage 4 (see $OUT/4_bench_gN.err)"; tail -40 $OUT/4_bench_gN.err; exit 4; }
cut -c1-400 $OUT/4_bench_gN.json
GPMI_DIST_COMM=rccl timeout -k 10 560 python3 bench.py --gpus $NG --steps 3 --warmup 1 > $OUT/4b_bench_gN_cabi.json 2> $OUT/4b_bench_gN_cabi.err \
    && cut -c1-400 $OUT/4b_bench_gN_cabi.json || say "the C-ABI backend at full size failed (see $OUT/4b_bench_gN_cabi.err)"
GPMI_BENCH_TWO_CALLS=1 timeout -k 10 560 python3 bench.py --gpus $NG --steps 3 --warmup 1 > $OUT/4c_bench_gN_two_calls.json 2> $OUT/4c_bench_gN_two_calls.err \
    && cut -c1-400 $OUT/4c_bench_gN_two_calls.json || say "the two-call step (A/B of the one-pass step; column-wise predict with its n x nb broadcasts) failed (see $OUT/4c_bench_gN_two_calls.err)"
say "all four stages passed"
