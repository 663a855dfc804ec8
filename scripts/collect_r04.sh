#!/bin/bash
# Round-4 evidence for profiles/ at the commit it is run on: the bench line (with extra_configs, measured CPU headline,
# transfer-inclusive step), kernel-trace stats of the bench command, separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy
# counters) over one step -- never combined with --stats / trace domains beyond --kernel-trace --, and the replays of
# every rank of 8 / 4 / 2.  Run on the GPU box from the repository root:
#   bash scripts/collect_r04.sh <tag>        (writes gpurun_out/<tag>_*; copy what is to be judged into profiles/)
# COLLECT_ONLY=trace: the kernel-trace and PMC passes alone.  Those passes run bench.py --no-other-form, so every launch of
# the trailing-update kernel in them belongs to the timed call form (the trace's average = the bench line's avg_launch_ms).
set -o pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rm -rf $OUT/prof_tmp && mkdir -p $OUT/prof_tmp
if [ "$COLLECT_ONLY" != "trace" ]; then
python3 bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench_N65536.json 2> $OUT/${TAG}_bench.err || { tail -20 $OUT/${TAG}_bench.err; exit 1; }
echo "bench done"
fi
rocprofv3 --kernel-trace --stats -d $OUT/prof_tmp/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-other-form > $OUT/${TAG}_bench_under_trace.json 2> $OUT/${TAG}_kt.err || exit 1
python3 scripts/rocpd_extract.py stats $(find $OUT/prof_tmp/kt -name "*.db" | head -1) $OUT/${TAG}_bench_N65536_kernel_stats.csv > $OUT/${TAG}_kernel_stats_top.txt || exit 1
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_tmp/f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --no-other-form > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_f.err || exit 1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_tmp/w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --no-other-form > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_w.err || exit 1
echo "write pass done"
python3 scripts/rocpd_extract.py traffic $(find $OUT/prof_tmp/f -name "*.db" | head -1) $(find $OUT/prof_tmp/w -name "*.db" | head -1) $OUT/${TAG}_roofline_traffic.json > /dev/null || exit 1
python3 scripts/rocpd_extract.py pmc $(find $OUT/prof_tmp/f -name "*.db" | head -1) $OUT/${TAG}_pmc_fetch_summary.txt > /dev/null
python3 scripts/rocpd_extract.py pmc $(find $OUT/prof_tmp/w -name "*.db" | head -1) $OUT/${TAG}_pmc_write_summary.txt > /dev/null
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $OUT/prof_tmp/s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --no-other-form > $OUT/${TAG}_pmc_sq.json 2> $OUT/${TAG}_s.err || exit 1
python3 scripts/rocpd_extract.py pmc $(find $OUT/prof_tmp/s -name "*.db" | head -1) $OUT/${TAG}_pmc_sq_summary.txt > /dev/null
echo "sq pass done"
rm -rf $OUT/prof_tmp
[ "$COLLECT_ONLY" = "trace" ] && { cat $OUT/${TAG}_kernel_stats_top.txt | head -12; cat $OUT/${TAG}_roofline_traffic.json; exit 0; }
for G in 8 4 2; do          # every rank of the G-rank run, one after the other (default layout and call form)
  RANKS=$(seq -s, 0 $((G-1)))
  python3 bench.py --replay-rank $RANKS --of $G --steps 3 --warmup 1 > $OUT/${TAG}_replay_G$G.json 2> $OUT/${TAG}_replay_G$G.err || exit 1
done
echo "replays done"
cat $OUT/${TAG}_kernel_stats_top.txt | head -12; cat $OUT/${TAG}_roofline_traffic.json
