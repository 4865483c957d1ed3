O=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $O
timeout -k 10 500 python scripts/train_glue.py > $O/glue.txt 2> $O/glue.err; tail -3 $O/glue.err; head -80 $O/glue.txt
