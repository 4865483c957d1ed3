O=$GRAFT_REPO_ROOT/gpurun_out/r03y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step.jsonl 2> $O/err.log || exit 1
TGP_SCATTER_FREE=0 python3 $R/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step_atomic_scatter.jsonl 2>> $O/err.log || exit 1
TGP_TRAIN_FACTORED=0 python3 $R/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step_unfactored.jsonl 2>> $O/err.log || exit 1
python3 $R/bench.py --workload train_step --batch 256 --steps 4 --warmup 2 --no-cpu-baseline > $O/b256_train.jsonl 2>> $O/err.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --workload train_step --steps 20 --warmup 3 --graph 0 --no-cpu-baseline --min-seconds 0 > $O/train_eager_under_rocprof.jsonl 2>> $O/err.log
find $O -name "*kernel_trace.csv" -delete
python3 -c "
import json
for f in ('train_step','train_step_atomic_scatter','train_step_unfactored','b256_train'):
    d=json.loads(open('$O/'+f+'.jsonl').read().strip().splitlines()[-1]);print(f, d['ms_per_step'],d['value'])"
