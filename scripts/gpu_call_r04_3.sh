#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_replay.py tests/test_abi.py tests/test_parity_gpu.py -x -q -m gpu -k "replay or abi or gives_up or misaligned or backward_solve or gemm or persist" > gpurun_out/r04_3_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_3_pytest.txt
timeout -k 10 600 python3 -m pytest tests/test_dist.py -x -q -m gpu > gpurun_out/r04_3_pytest_dist.txt 2>&1; echo "pytest dist rc=$?"; tail -5 gpurun_out/r04_3_pytest_dist.txt
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from gaussian_process_amd import GPContext
with GPContext(0) as ctx:
    for (M, N, K, lower) in ((8192, 8192, 512, 1), (8192, 8192, 1024, 1), (4096, 4096, 512, 1), (12288, 12288, 1024, 1), (16384, 16384, 2048, 1), (32768, 32768, 2048, 1), (8192, 61440, 1024, 0), (2048, 16384, 1024, 0)):
        for name, pers, tk in (("tile", 0, 0), ("persist+steal", 1, 0), ("ticket", 0, 2)):
            ctx.set_option("gemm_persist", pers); ctx.set_option("gemm_ticket", tk)
            tf, ms = ctx.probe_gemm(M, N, K, lower, 0, 5)
            print("probe %dx%dx%d lower=%d %s: %.2f TF/s %.3f ms" % (M, N, K, lower, name, tf, ms), flush=True)
PY
python3 scripts/small_sizes.py 2>&1 | grep -v amdgpu.ids
for LAYOUT in snake cyclic; do
for G in 8 4 2; do
  GPMI_DIST_LAYOUT=$LAYOUT GPMI_REPLAY_LAYOUT=$LAYOUT timeout -k 10 300 python3 bench.py --replay-rank 0,$((G-1)) --of $G --steps 3 --warmup 1 > gpurun_out/r04_replay_${LAYOUT}_G$G.json 2> gpurun_out/r04_replay_${LAYOUT}_G$G.err; echo "replay $LAYOUT G=$G rc=$?"
  python3 -c "
import json
j=json.load(open('gpurun_out/r04_replay_${LAYOUT}_G$G.json'))
print('$LAYOUT G=%d nb=%d t1=%.1f worst=%.1f bound=%.2f' % (j['of'], j['block_rows'], j['t1_ms'], j['worst_rank_ms'], j['speedup_upper_bound']))
for r in j['ranks']:
    d=r['diag']
    print('   rank %d: %.1f ms fit %.1f alpha %.1f predict %.1f | update %.1f stall %.1f panel_solve %.1f diag %.1f host %.1f | L_rel %.1e lml_rel %.1e' % (r['rank'], r['ms_per_step'], r['fit_ms'], r['alpha_ms'], r['predict_ms'], d['update_ms'], d['stall_panel_ms'], d['panel_solve_ms'], d.get('diag_ms',0), d['host_issue_ms'], r['L_rel'], r['lml_rel_vs_source']))
" || tail -20 gpurun_out/r04_replay_${LAYOUT}_G$G.err
done
done
