O=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reverse_graph or gather_vs_scatter or hs_layer_backward or pool_and_upsample or backward_encoder_only" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step.jsonl 2> $O/err0.log || exit 1
TGP_SCATTER_FREE=0 python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step_atomics.jsonl 2>> $O/err0.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 20 --warmup 3 --graph 0 --no-cpu-baseline --min-seconds 0 > $O/train_eager.jsonl 2> $O/err.log
TGP_SCATTER_FREE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_atomics -- python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 20 --warmup 3 --graph 0 --no-cpu-baseline --min-seconds 0 > $O/train_eager_atomics.jsonl 2> $O/err.log
find $O -name "*kernel_trace.csv" -delete
python3 -c "
import json
for f in ('train_step','train_step_atomics'):
    d=json.loads(open('$O/'+f+'.jsonl').read().strip().splitlines()[-1]);print(f, d['ms_per_step'],d['value'])"
