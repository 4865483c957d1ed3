mkdir -p gpurun_out/r03p
export TMPDIR=/tmp
for f in 1 0 1 0; do
  TGP_TRAIN_FACTORED=$f timeout -k 10 300 python bench.py --workload train_step --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('factored=$f', d['ms_per_step'], d['value'], d['config'])" >> gpurun_out/r03p/ab.log || exit 1
done
cat gpurun_out/r03p/ab.log
