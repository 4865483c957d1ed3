O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bn_ or backward or train or captured or graphed or feat_consumers" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step.jsonl 2> $O/err0.log || exit 1
python3 -c "
import json
for f in ('train_step',):
    d=json.loads(open('$O/'+f+'.jsonl').read().strip().splitlines()[-1]);print(f, d['ms_per_step'],d['value'])"
