#!/bin/bash
# Evidence run on the GPU box (one gpurun call): bench lines, rocprofv3 kernel summaries, FETCH / WRITE traffic and SQ counters of the
# eval forward on the current code.  Usage: bash scripts/collect_profiles.sh <tag>     (writes gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r03x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SERIAL="--steps 4 --warmup 2 --no-cpu-baseline --graph 0 --streams 1 --no-branch-streams --min-seconds 0"
python3 $R/bench.py > $O/bench.jsonl 2> $O/bench.err
python3 $R/bench.py --no-cpu-baseline --streams 1 > $O/bench_one_in_flight.jsonl 2>> $O/bench.err
python3 $R/bench.py --no-cpu-baseline --streams 2 > $O/bench_two_in_flight.jsonl 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.jsonl 2>> $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --graph 0 --streams 1 --no-branch-streams --min-seconds 0 > $O/bench_serial_eager.jsonl 2>> $O/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $SERIAL > /dev/null 2>> $O/prof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $SERIAL > /dev/null 2>> $O/prof.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 $R/bench.py $SERIAL > /dev/null 2>> $O/prof.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py $SERIAL > /dev/null 2>> $O/prof.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq3 -- python3 $R/bench.py $SERIAL > /dev/null 2>> $O/prof.err
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null 2>> $O/prof.err
python3 $R/scripts/sq_counters.py $O/sq_counters_forward.txt $O/sq1 $O/sq2 $O/sq3 > /dev/null 2>> $O/prof.err
python3 $R/scripts/forward_launches.py $(find $O/prof_serial -name "*kernel_trace.csv" | head -1) $O/forward_launches_one_forward.txt >> $O/prof.err 2>&1
python3 $R/bench.py --no-cpu-baseline --eval-outputs-only > $O/bench_eval_outputs_only.jsonl 2>> $O/bench.err
python3 $R/bench.py --no-cpu-baseline --workload train_step > $O/train_step.jsonl 2>> $O/bench.err
python3 $R/bench.py --no-cpu-baseline --workload train_step --batch 256 --steps 5 --warmup 2 > $O/b256_train.jsonl 2>> $O/bench.err
# keep the merge small: summaries only
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
echo collected
