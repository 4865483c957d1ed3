O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tight_on_flip_free" -s > $O/t.log 2>&1; tail -12 $O/t.log
