"""Split-GEMM launch durations of a rocprofv3 kernel trace of the default `bench.py` command, by phase of the run.
The default bench keeps two graph replays in flight during the timed region, which stretches every kernel's duration in the
trace (two kernels share the CUs); the roofline block is measured on the serial eager steps that follow.  This script cuts
the trace at those boundaries so the summary can be compared with `roofline.avg_launch_us`.
usage: python scripts/trace_phases.py <kernel_trace.csv> [steps=20] [warmup=5] [streams=2] [launches_per_forward=14]"""
import csv
import sys

path = sys.argv[1]
steps, warmup, streams, per = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 20), (3, 5), (4, 2), (5, 14)))
rows = csv.DictReader(open(path))
a = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
           if "gemm_split" in r["Kernel_Name"] and "_kernel" in r["Kernel_Name"])
cuts = [("graph construction (2 warm runs per captured forward)", 2 * streams * per),
        ("warm-up replays, %d in flight" % streams, max(warmup, streams) * per),
        ("TIMED replays, %d in flight" % streams, steps * per),
        ("eager warm step", per),
        ("serial eager steps of the roofline block", min(steps, 10) * per),
        ("replays with one batch in flight", steps * per)]
at = 0
print("%d split-GEMM launches in the trace" % len(a))
for name, n in cuts:
    seg = a[at:at + n]
    at += n
    if seg:
        print("%-62s %5d launches  mean %7.1f us" % (name, len(seg), sum(x[1] for x in seg) / len(seg) / 1e3))
if at != len(a):
    print("(%d launches not attributed: the phase sizes above assume the default flags)" % (len(a) - at))
