#!/bin/bash
# the one-pass step in the bench line + the whole GPU suite
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 bench.py --steps 6 --warmup 2 > gpurun_out/r04_45_bench.json 2> gpurun_out/r04_45_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_45_bench.json'))
print(d['value'], d['ms_per_step'], d.get('seconds_incl_transfers'))
print(d['config'].get('call_form'))
print(d.get('other_call_form'))
print({k:d['roofline'][k] for k in ('achieved','frac','avg_launch_ms','launches_per_step','flops_per_launch')})
print(d.get('solve_v_mfma'))
print(d.get('stages_ms'))
for k,v in d.get('extra_configs',{}).items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in('ms_per_step','ms_per_step_two_calls','seconds','tflops')})
PY
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_45_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_45_pytest_gpu.txt
