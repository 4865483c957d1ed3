set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_train; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --no-cpu-baseline --workload train_step --graph 0 --steps 6 --warmup 3 --min-seconds 0 > $O/train_eager.jsonl 2> $O/err.txt &&
python3 $R/bench.py --no-cpu-baseline --workload train_step > $O/train.jsonl 2>> $O/err.txt
find $O -name "*kernel_trace.csv" -size +30M -delete
echo done
