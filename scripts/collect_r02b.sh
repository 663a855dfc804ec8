#!/bin/bash
# Refresh of the round-2 evidence after the third-generation backward solve: the bench line, the kernel-trace stats of
# the bench command and the forced-dist run (PMC passes of the trailing update are unchanged: same kernel).
#   bash scripts/collect_r02b.sh <tag>        (writes gpurun_out/<tag>_*)
set -o pipefail
TAG=${1:-r02b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rm -rf $OUT/prof_tmp && mkdir -p $OUT/prof_tmp
python3 bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench_N65536.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench done"
GPMI_BENCH_FORCE_DIST=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_forced_dist.json 2> $OUT/${TAG}_fd.err || exit 1
echo "forced dist done"
rocprofv3 --kernel-trace --stats -d $OUT/prof_tmp/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_under_trace.json 2> $OUT/${TAG}_kt.err || exit 1
python3 scripts/rocpd_extract.py stats $(find $OUT/prof_tmp/kt -name "*.db" | head -1) $OUT/${TAG}_bench_N65536_kernel_stats.csv > $OUT/${TAG}_kernel_stats_top.txt || exit 1
echo "kernel trace done"
rm -rf $OUT/prof_tmp
head -12 $OUT/${TAG}_kernel_stats_top.txt
