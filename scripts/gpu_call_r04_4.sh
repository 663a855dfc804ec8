#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for TK in 1 0; do
  GPMI_DIST_TICKET=$TK GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,3,7 --of 8 --steps 3 --warmup 1 > gpurun_out/r04_replay_ticket${TK}_G8.json 2> gpurun_out/r04_replay_ticket${TK}_G8.err; echo "replay ticket=$TK rc=$?"
  python3 -c "
import json
j=json.load(open('gpurun_out/r04_replay_ticket${TK}_G8.json'))
for r in j['ranks']:
    d=r['diag']
    print('ticket=$TK rank %d: %.1f ms fit %.1f alpha %.1f predict %.1f | update %.1f stall %.1f panel_solve %.1f diag %.1f host %.1f update_v %.1f| L_rel %.1e lml_rel %.1e' % (r['rank'], r['ms_per_step'], r['fit_ms'], r['alpha_ms'], r['predict_ms'], d['update_ms'], d['stall_panel_ms'], d['panel_solve_ms'], d.get('diag_ms',0), d['host_issue_ms'], d['update_v_ms'], r['L_rel'], r['lml_rel_vs_source']))
" || tail -20 gpurun_out/r04_replay_ticket${TK}_G8.err
done
python3 scripts/opt_combo.py 16384 1024 "" "panel_prio=1" "gemm_stagger=2" "gemm_stagger=4" "panel_prio=1,gemm_stagger=2" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_4_opt_16384.txt
python3 scripts/opt_combo.py 32768 4096 "" "panel_prio=1" "gemm_stagger=2" "gemm_stagger=4" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_4_opt_32768.txt
python3 scripts/opt_combo.py 65536 4096 "" "panel_prio=1" "gemm_stagger=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_4_opt_65536.txt
