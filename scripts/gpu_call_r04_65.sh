#!/bin/bash
# end of round: the new test, then the bench line at the final commit
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "lookahead_threshold or lanes" > gpurun_out/r04_65_pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r04_65_pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench.py --steps 10 --warmup 2 > gpurun_out/r04d_bench_N65536.json 2> gpurun_out/r04d_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04d_bench_N65536.json'))
print(d['value'], d['ms_per_step'], d.get('seconds_incl_transfers'), d['other_call_form']['ms_per_step'])
print({k:d['roofline'][k] for k in ('achieved','frac','avg_launch_ms','traffic')}, d['roofline']['traffic_source'])
for k,v in d.get('extra_configs',{}).items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in('ms_per_step','ms_per_step_two_calls','seconds','tflops')})
PY
