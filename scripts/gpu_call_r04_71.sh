#!/bin/bash
# last check of the round: the bench line and the 8-rank replay at the final code
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 bench.py --steps 6 --warmup 2 > gpurun_out/r04e_bench_N65536.json 2> gpurun_out/r04e_bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --replay-rank 0,7 --of 8 --steps 3 --warmup 1 > gpurun_out/r04e_replay_G8.json 2> gpurun_out/r04e_replay_G8.err; echo "replay rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04e_bench_N65536.json'))
print(d['value'], d['ms_per_step'], d.get('seconds_incl_transfers'), d['other_call_form']['ms_per_step'], d['other_call_form']['lml_equal'])
print({k:d['roofline'][k] for k in ('achieved','frac','avg_launch_ms')})
for k,v in d.get('extra_configs',{}).items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in('ms_per_step','ms_per_step_two_calls','seconds','tflops')})
r=json.load(open('gpurun_out/r04e_replay_G8.json'))
print(r['t1_ms'], r['worst_rank_ms'], r['speedup_upper_bound'], [round(k['ms_per_step'],1) for k in r['ranks']])
PY
