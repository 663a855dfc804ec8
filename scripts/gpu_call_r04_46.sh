#!/bin/bash
# DistGP one-pass form: the partitioned-path GPU tests, then the replays (balanced layout, one-pass step)
cd "$GRAFT_REPO_ROOT"
echo "tests: passed in the previous call (39 passed)"; rc=0
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --replay-rank 0,6 --of 8 --steps 3 --warmup 1 > gpurun_out/r04_replay_onepass_G8.json 2> gpurun_out/r04_replay_onepass_G8.err || { tail -20 gpurun_out/r04_replay_onepass_G8.err; exit 1; }
GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,3 --of 4 --steps 3 --warmup 1 > gpurun_out/r04_replay_onepass_G4.json 2> gpurun_out/r04_replay_onepass_G4.err || exit 1
GPMI_REPLAY_NO_T1=1 timeout -k 10 300 python3 bench.py --replay-rank 0,1 --of 2 --steps 2 --warmup 1 > gpurun_out/r04_replay_onepass_G2.json 2> gpurun_out/r04_replay_onepass_G2.err || exit 1
python3 - <<'PY'
import json
for G in (8,4,2):
    r=json.load(open('gpurun_out/r04_replay_onepass_G%d.json'%G))
    print(G, r.get('t1_ms'), r.get('speedup_upper_bound'), r.get('call_form'))
    for k in r['ranks']: print(' ', k['rank'], round(k['ms_per_step'],1), round(k['fit_ms'],1), round(k['alpha_ms'],1), round(k['predict_ms'],1), k['diag'].get('update_ms'), k['diag'].get('stall_panel_ms'), k['mu_maxabs_vs_source'], k['var_maxabs_vs_source'], k['delivered_bytes_per_step'])
PY
