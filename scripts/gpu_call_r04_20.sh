#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for LA in 0 1 2; do
GPMI_DIST_LOOKAHEAD=$LA GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0 --of 8 --steps 2 --warmup 1 > gpurun_out/t_la$LA.json 2> gpurun_out/t_la$LA.err
python3 -c "
import json
j=json.load(open('gpurun_out/t_la$LA.json'))
for r in j['ranks']:
    d=r['diag']
    print('lookahead=$LA rank %d: %.1f ms fit %.1f predict %.1f | update %.1f (n=%d) stall %.1f panel_solve %.1f diag %.1f pack %.1f allgather %.1f update_v %.1f solve_v %.1f' % (r['rank'], r['ms_per_step'], r['fit_ms'], r['predict_ms'], d['update_ms'], d['update_n'], d.get('stall_panel_ms',0), d['panel_solve_ms'], d.get('diag_ms',0), d['pack_ms'], d['allgather_ms'], d['update_v_ms'], d['solve_v_ms']))
"
done
