#!/bin/bash
# GPU box: the default bench line under different environment switches, alternating on the SAME box (boxes differ by +-3 %).
# Usage: bash scripts/env_ab.sh <rounds> "ENV_A=.." "ENV_B=.." ...
cd /tmp
R=$GRAFT_REPO_ROOT
ROUNDS=$1; shift
for rep in $(seq 1 $ROUNDS); do
  for e in "$@"; do
    env $e python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s four in flight %8.1f   one %8.1f   two %8.1f' % ('$e', d['value'], d['config'].get('objects_per_s_one_batch_in_flight'), d['config'].get('objects_per_s_two_batches_in_flight')))"
  done
done
