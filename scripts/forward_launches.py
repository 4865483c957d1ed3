"""One eval forward's kernel launches, in order, from a rocprofv3 kernel trace of serial eager forwards
(`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --graph 0 --streams 1
--no-branch-streams --min-seconds 0`): the last-but-one forward of the trace, from its center_mean_kernel to the launch before the next forward's.
usage: python scripts/forward_launches.py <kernel_trace.csv> [out.txt]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("center_mean_kernel")]
i0, i1 = starts[-2], starts[-1] - 1            # the last forward that is followed by another one: up to the next forward's first launch
fwd = rows[i0:i1 + 1]
t0 = int(fwd[0]["Start_Timestamp"])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(" start us   dur us  kernel  [grid x workgroup]", file=out)
tot = 0.0
for r in fwd:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    print("%9.1f %8.1f  %s  [%s x %s]" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, name[:90], r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                                           r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))), file=out)
print("\nlaunches: %d   kernel time: %.1f us" % (len(fwd), tot), file=out)
