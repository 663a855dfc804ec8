cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_tmp && mkdir -p gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp/kt -- python3 scripts/fused_ab.py 16384 > gpurun_out/r2c_trace16k.out 2> gpurun_out/r2c_trace16k.err || exit 1
python3 scripts/rocpd_extract.py stats $(find gpurun_out/prof_tmp/kt -name "*.db" | head -1) gpurun_out/r2c_trace16k_stats.csv > gpurun_out/r2c_trace16k_top.txt
rm -rf gpurun_out/prof_tmp
head -16 gpurun_out/r2c_trace16k_top.txt
