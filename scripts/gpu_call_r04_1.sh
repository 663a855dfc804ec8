#!/bin/bash
# round 4, first GPU call: new tests, ticket A/B, replays
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_replay.py tests/test_abi.py -x -q -m gpu > gpurun_out/r04_1_pytest_replay.txt 2>&1; echo "pytest replay rc=$?"; tail -3 gpurun_out/r04_1_pytest_replay.txt
timeout -k 10 400 python3 -m pytest tests/test_dist.py -x -q -m gpu -k "eight_ranks_on_one_gpu" > gpurun_out/r04_1_pytest_dist.txt 2>&1; echo "pytest dist rc=$?"; tail -3 gpurun_out/r04_1_pytest_dist.txt
timeout -k 10 400 python3 scripts/ticket_ab.py > gpurun_out/r04_1_ticket_ab.txt 2>&1; echo "ticket rc=$?"; tail -40 gpurun_out/r04_1_ticket_ab.txt
for G in 8 4 2; do
  timeout -k 10 300 python3 bench.py --replay-rank 0,$((G-1)) --of $G --steps 3 --warmup 1 > gpurun_out/r04_replay_G$G.json 2> gpurun_out/r04_replay_G$G.err; echo "replay G=$G rc=$?"
  python3 -c "
import json,sys
j=json.load(open('gpurun_out/r04_replay_G$G.json'))
print('G=%d nb=%d t1=%.1f worst=%.1f bound=%.2f' % (j['of'], j['block_rows'], j['t1_ms'], j['worst_rank_ms'], j['speedup_upper_bound']))
for r in j['ranks']:
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k!='diag'})
    print('   diag', r['diag'])
" || tail -20 gpurun_out/r04_replay_G$G.err
done
