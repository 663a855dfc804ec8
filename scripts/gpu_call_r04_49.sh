#!/bin/bash
# config 4's size (N=131072, d=16) as one rank of 8, one-pass step, balanced layout
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python3 bench.py --size 131072 --dim 16 --replay-rank 0,7 --of 8 --steps 1 --warmup 1 > gpurun_out/r04_replay_onepass_G8_N131072_d16.json 2> gpurun_out/r04_replay_onepass_G8_N131072_d16.err || { tail -20 gpurun_out/r04_replay_onepass_G8_N131072_d16.err; exit 1; }
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/r04_replay_onepass_G8_N131072_d16.json'))
print(r.get('t1_ms'), r.get('speedup_upper_bound'), r.get('block_rows'), r.get('call_form'))
for k in r['ranks']: print(' ', k['rank'], round(k['ms_per_step'],1), round(k['fit_ms'],1), k['diag'].get('update_ms'), k['diag'].get('stall_panel_ms'), k['mu_maxabs_vs_source'])
PY
