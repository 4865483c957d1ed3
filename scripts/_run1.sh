O=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.jsonl 2> $O/err.log || exit 1
python3 $R/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline > $O/train_step.jsonl 2>> $O/err.log || exit 1
python3 $R/bench.py --workload train_step --batch 256 --steps 4 --warmup 2 --no-cpu-baseline > $O/b256_train.jsonl 2>> $O/err.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --workload train_step --steps 20 --warmup 3 --graph 0 --no-cpu-baseline --min-seconds 0 > $O/train_eager_under_rocprof.jsonl 2>> $O/err.log
find $O -name "*kernel_trace.csv" -delete
cd $R
TGP_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 6 --warmup 2 --workload train_step --no-cpu-baseline > $O/r2_train.jsonl 2>> $O/err.log || exit 1
python3 -c "
import json
for f in ('bench','train_step','b256_train','r2_train'):
    d=json.loads(open('$O/'+f+'.jsonl').read().strip().splitlines()[-1]);print(f, d['ms_per_step'],d['value'])"
