O=$GRAFT_REPO_ROOT/gpurun_out/r03w
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "eval_outputs_only or forward_vs or full_forward" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
python3 bench.py --no-cpu-baseline > $O/bench_full.jsonl 2> $O/err0.log || exit 1
python3 bench.py --no-cpu-baseline --eval-outputs-only > $O/bench_eval_outputs_only.jsonl 2>> $O/err0.log || exit 1
python3 bench.py --no-cpu-baseline --eval-outputs-only --streams 1 > $O/bench_eval_outputs_only_one.jsonl 2>> $O/err0.log || exit 1
python3 -c "
import json
for f in ('bench_full','bench_eval_outputs_only','bench_eval_outputs_only_one'):
    d=json.loads(open('$O/'+f+'.jsonl').read().strip().splitlines()[-1]);print(f, d['ms_per_step'],d['value'], d['roofline']['frac'])"
