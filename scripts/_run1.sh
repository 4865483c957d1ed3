O=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 600 python3 scripts/eval_pipeline.py > $O/eval_pipeline.txt 2>> $O/err.log || exit 1
python3 -c "
import json
for l in open('$O/eval_pipeline.txt'):
    if l.startswith('{'):
        d=json.loads(l); print('lean', d['sampler'], d['hipgraph'], d['frames_per_s'], d['objects_per_s'])"
TGP_EVAL_LEAN_OFF=1 timeout -k 10 600 python3 scripts/eval_pipeline.py > $O/eval_pipeline_full.txt 2>> $O/err.log || exit 1
python3 -c "
import json
for l in open('$O/eval_pipeline_full.txt'):
    if l.startswith('{'):
        d=json.loads(l); print('full', d['sampler'], d['hipgraph'], d['frames_per_s'], d['objects_per_s'])"
done
