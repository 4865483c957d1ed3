O=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_repeatable" > $O/t.log 2>&1; tail -15 $O/t.log
