O=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $O
timeout -k 10 500 python scripts/train_glue_phases.py > $O/phases.txt 2> $O/phases.err; tail -3 $O/phases.err; cat $O/phases.txt
