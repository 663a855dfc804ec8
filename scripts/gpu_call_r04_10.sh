#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
show() { python3 -c "
import json,sys
j=json.load(open(sys.argv[1]))
for r in j['ranks']:
    d=r['diag']
    print(sys.argv[2], 'rank %d: %.1f ms fit %.1f alpha %.1f predict %.1f | update %.1f stall %.1f panel_solve %.1f host %.1f update_v %.1f' % (r['rank'], r['ms_per_step'], r['fit_ms'], r['alpha_ms'], r['predict_ms'], d['update_ms'], d['stall_panel_ms'], d['panel_solve_ms'], d['host_issue_ms'], d['update_v_ms']))
" $1 "$2"; }
timeout -k 10 300 python3 bench.py --replay-rank 0,7 --of 8 --steps 3 --warmup 1 > gpurun_out/t_a.json 2> gpurun_out/t_a.err; show gpurun_out/t_a.json "with-T1 order 0,7:"
GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,7 --of 8 --steps 3 --warmup 1 > gpurun_out/t_b.json 2> gpurun_out/t_b.err; show gpurun_out/t_b.json "no-T1 order 0,7:"
GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 7,0 --of 8 --steps 3 --warmup 1 > gpurun_out/t_c.json 2> gpurun_out/t_c.err; show gpurun_out/t_c.json "no-T1 order 7,0:"
GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,0,4 --of 8 --steps 3 --warmup 1 > gpurun_out/t_d.json 2> gpurun_out/t_d.err; show gpurun_out/t_d.json "no-T1 order 0,0,4:"
