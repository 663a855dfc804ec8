#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --steps 5 --warmup 1 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc=$?"
python3 -c "
import json
j=json.load(open('gpurun_out/r04_bench_default.json'))
print('value %.2f TF/s  ms_per_step %.1f  incl transfers %.4f s' % (j['value'], j['ms_per_step'], j.get('seconds_incl_transfers', -1)))
print('roofline', {k: j['roofline'][k] for k in ('achieved','frac','avg_launch_ms','traffic')})
print('extra', json.dumps(j.get('extra_configs'), indent=1)[:1500])
print('cpu', {k: j['cpu_baseline'][k] for k in ('value','cores','seconds')}, j['cpu_baseline'].get('measured_headline',{}).get('seconds'))
print('kbuild', j['kbuild_hbm']['frac'], j['kbuild_hbm'].get('valu_count_source'))
" || tail -20 gpurun_out/r04_bench_default.err
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_6_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r04_6_pytest_gpu.txt
