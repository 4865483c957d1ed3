#!/bin/bash
# GPU box: the default bench line, one batch in flight, and the launch list of one serial eager forward.  Usage: bash scripts/quick_bench.sh <tag>
set -o pipefail
TAG=${1:-q}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline > $O/bench.jsonl 2> $O/bench.err
python3 $R/bench.py --no-cpu-baseline --streams 1 > $O/bench_one_in_flight.jsonl 2>> $O/bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --graph 0 --streams 1 --no-branch-streams --min-seconds 0 > $O/bench_serial_eager.jsonl 2>> $O/prof.err
python3 $R/scripts/forward_launches.py $(find $O/prof_serial -name "*kernel_trace.csv" | head -1) $O/forward_launches.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import json
for f in ("bench.jsonl", "bench_one_in_flight.jsonl"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
PY
tail -2 $O/forward_launches.txt
