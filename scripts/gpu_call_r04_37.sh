#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for BAL in 1 0 1 0; do
  GPMI_DIST_BALANCE=$BAL GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,7 --of 8 --steps 3 --warmup 1 > gpurun_out/r04_replay_sebal${BAL}_G8.json 2> gpurun_out/r04_replay_sebal${BAL}_G8.err; echo "replay balance=$BAL rc=$?"
  python3 -c "
import json
j=json.load(open('gpurun_out/r04_replay_sebal${BAL}_G8.json'))
for r in j['ranks']:
    d=r['diag']
    print('balance=$BAL rank %d: %.1f ms fit %.1f alpha %.1f predict %.1f | update %.1f stall %.1f panel_solve %.1f host %.1f update_v %.1f | L_rel %.1e' % (r['rank'], r['ms_per_step'], r['fit_ms'], r['alpha_ms'], r['predict_ms'], d['update_ms'], d['stall_panel_ms'], d['panel_solve_ms'], d['host_issue_ms'], d['update_v_ms'], r['L_rel']))
" || tail -20 gpurun_out/r04_replay_sebal${BAL}_G8.err
done
timeout -k 10 400 python3 -m pytest tests/test_dist.py tests/test_replay.py -x -q -m gpu 2>&1 | tail -3
