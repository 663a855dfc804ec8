#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for NB in 512 2048; do
  GPMI_DIST_NB=$NB GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,7 --of 8 --steps 2 --warmup 1 > gpurun_out/r04_replay_G8_nb$NB.json 2> gpurun_out/r04_replay_G8_nb$NB.err; echo "replay nb=$NB rc=$?"
  python3 -c "
import json
j=json.load(open('gpurun_out/r04_replay_G8_nb$NB.json'))
for r in j['ranks']:
    d=r['diag']
    print('nb=%d rank %d: %.1f ms fit %.1f alpha %.1f predict %.1f | update %.1f stall %.1f panel_solve %.1f diag %.1f host %.1f' % (j['block_rows'], r['rank'], r['ms_per_step'], r['fit_ms'], r['alpha_ms'], r['predict_ms'], d['update_ms'], d['stall_panel_ms'], d['panel_solve_ms'], d['diag_ms'], d['host_issue_ms']))
"
done
# persist vs ticket, standalone
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from gaussian_process_amd import GPContext
with GPContext(0) as ctx:
    for (M, N, K, lower) in ((8192, 8192, 512, 1), (8192, 8192, 1024, 1), (4096, 4096, 512, 1), (12288, 12288, 1024, 1), (16384, 16384, 2048, 1), (8192, 61440, 1024, 0), (4096, 30720, 1024, 0), (2048, 16384, 1024, 0)):
        for name, pers, tk in (("tile", 0, 0), ("persist", 1, 0), ("ticket", 0, 2)):
            ctx.set_option("gemm_persist", pers); ctx.set_option("gemm_ticket", tk)
            tf, ms = ctx.probe_gemm(M, N, K, lower, 0, 5)
            print("probe %dx%dx%d lower=%d %s: %.2f TF/s %.3f ms" % (M, N, K, lower, name, tf, ms), flush=True)
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_tmp && mkdir -p gpurun_out/prof_tmp
GPMI_REPLAY_NO_T1=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp/kt -- python3 bench.py --replay-rank 7 --of 8 --steps 1 --warmup 1 > gpurun_out/r04_replay_trace.json 2> gpurun_out/r04_replay_trace.err; echo "trace rc=$?"
DB=$(find gpurun_out/prof_tmp/kt -name "*.db" | head -1)
python3 scripts/rocpd_extract.py stats $DB gpurun_out/r04_replay_rank7_kernel_stats.csv > gpurun_out/r04_replay_rank7_kernel_stats_top.txt; head -30 gpurun_out/r04_replay_rank7_kernel_stats_top.txt
python3 scripts/trace_timeline.py $DB > gpurun_out/r04_replay_rank7_timeline.txt 2>&1; head -60 gpurun_out/r04_replay_rank7_timeline.txt
rm -rf gpurun_out/prof_tmp
