#!/bin/bash
# the GPU suite again (one tolerance fixed) + the "balanced" row-block layout in the replay (ranks: last-block owner 0, heaviest 6, lightest 3)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_42_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_42_pytest_gpu.txt
GPMI_DIST_LAYOUT=balanced timeout -k 10 300 python3 bench.py --replay-rank 0,6,3 --of 8 --steps 3 --warmup 1 > gpurun_out/r04_replay_balanced_G8.json 2> gpurun_out/r04_replay_balanced_G8.err || exit 1
GPMI_DIST_LAYOUT=balanced GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,3,1 --of 4 --steps 3 --warmup 1 > gpurun_out/r04_replay_balanced_G4.json 2> gpurun_out/r04_replay_balanced_G4.err || exit 1
python3 - <<'PY'
import json
for G in (8,4):
    r=json.load(open('gpurun_out/r04_replay_balanced_G%d.json'%G))
    print(G, r.get('t1_ms'), r.get('speedup_upper_bound'))
    for k in r['ranks']: print(' ', k['rank'], round(k['ms_per_step'],1), round(k['fit_ms'],1), k['diag']['update_ms'], k['diag']['stall_panel_ms'])
PY
