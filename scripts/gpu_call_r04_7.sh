#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_dist.py tests/test_abi.py tests/test_parity_gpu.py -x -q -m gpu -k "rccl or abi or ticket" > gpurun_out/r04_7_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04_7_pytest.txt
timeout -k 10 300 python3 scripts/nccl_world1.py 6144 rccl 2>&1 | grep -v amdgpu.ids | tail -6
# the multi-rank bench line through the C-ABI RCCL binding on a world of one (rehearsal) against torch's process group
GPMI_BENCH_FORCE_DIST=1 GPMI_DIST_COMM=rccl timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_forced_dist_cabi.json 2> gpurun_out/r04_bench_forced_dist_cabi.err; echo "forced dist (C-ABI rccl) rc=$?"
GPMI_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_forced_dist_torch.json 2> gpurun_out/r04_bench_forced_dist_torch.err; echo "forced dist (torch nccl) rc=$?"
python3 -c "
import json
for w in ('cabi','torch'):
    j=json.load(open('gpurun_out/r04_bench_forced_dist_%s.json' % w))
    print(w, 'ms_per_step %.1f' % j['ms_per_step'], j['stages_ms'], j['comm'].get('backend'), j['comm'].get('library'))
" || tail -20 gpurun_out/r04_bench_forced_dist_cabi.err
# config 4's size: ranks 0 and 7 of 8 replayed
timeout -k 10 500 python3 bench.py --replay-rank 0,7 --of 8 --size 131072 --dim 16 --steps 1 --warmup 1 > gpurun_out/r04_replay_snake_G8_N131072_d16.json 2> gpurun_out/r04_replay_cfg4.err; echo "replay cfg4 rc=$?"
python3 -c "
import json
j=json.load(open('gpurun_out/r04_replay_snake_G8_N131072_d16.json'))
print('cfg4 G=%d nb=%d t1=%.1f worst=%.1f bound=%.2f' % (j['of'], j['block_rows'], j['t1_ms'], j['worst_rank_ms'], j['speedup_upper_bound']))
for r in j['ranks']:
    d=r['diag']
    print('   rank %d: %.1f ms fit %.1f alpha %.1f predict %.1f | update %.1f stall %.1f host %.1f | L_rel %.1e lml_rel %.1e' % (r['rank'], r['ms_per_step'], r['fit_ms'], r['alpha_ms'], r['predict_ms'], d['update_ms'], d['stall_panel_ms'], d['host_issue_ms'], r['L_rel'], r['lml_rel_vs_source']))
" || tail -20 gpurun_out/r04_replay_cfg4.err
