cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rm -rf $OUT/prof_tmp && mkdir -p $OUT/prof_tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_tmp/v -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r03b_pmc_valu.json 2> $OUT/r03b_v.err || exit 1
python3 scripts/rocpd_extract.py pmc $(find $OUT/prof_tmp/v -name "*.db" | head -1) $OUT/r03b_pmc_valu_summary.txt > /dev/null
rm -rf $OUT/prof_tmp
grep -A5 "rbf_regs" $OUT/r03b_pmc_valu_summary.txt | head -8
