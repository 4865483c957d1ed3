O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train or captured or graphed or rccl or nan_step or overlapped" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
TGP_NET2_BESIDE=$v python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 30 --warmup 5 --no-cpu-baseline 2>> $O/err0.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beside=$v', d['ms_per_step'], d['value'])" || exit 1
done
