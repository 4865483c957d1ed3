O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bn_" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
