O=$GRAFT_REPO_ROOT/gpurun_out/r03x
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "exit $?" >> $O/full.log; tail -5 $O/full.log
