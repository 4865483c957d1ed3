O=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 20 --warmup 3 --graph 0 --no-cpu-baseline --min-seconds 0 > $O/train_eager.jsonl 2> $O/err.log
find $O -name "*kernel_trace.csv" -delete
