#!/bin/bash
# 8-rank replay with block rows 2048 (4 blocks per rank, balanced layout), every rank
cd "$GRAFT_REPO_ROOT"
GPMI_DIST_NB=2048 timeout -k 10 500 python3 bench.py --replay-rank 0,1,2,3,4,5,6,7 --of 8 --steps 3 --warmup 1 > gpurun_out/r04_replay_onepass_G8_nb2048.json 2> gpurun_out/r04_replay_onepass_G8_nb2048.err || { tail -20 gpurun_out/r04_replay_onepass_G8_nb2048.err; exit 1; }
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/r04_replay_onepass_G8_nb2048.json'))
print(r.get('t1_ms'), r.get('speedup_upper_bound'), r.get('block_rows'))
for k in r['ranks']: print(' ', k['rank'], round(k['ms_per_step'],1), round(k['fit_ms'],1), k['diag'].get('update_ms'), k['diag'].get('stall_panel_ms'), k['diag'].get('host_issue_ms'))
PY
