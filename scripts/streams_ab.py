"""Development (GPU box): batches in flight x side branches inside each captured forward (hipGraph runs a graph's extra branches on a
pool of internal streams shared by every graph in flight; a branch-free graph stays on the stream it is launched on)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for branch in (1, 0):
    for streams in (1, 2, 3, 4):
        args = ["--no-cpu-baseline", "--streams", str(streams), "--min-seconds", "0.5"] + ([] if branch else ["--no-branch-streams"])
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True).stdout.strip().splitlines()
        try:
            d = json.loads(out[-1])
            print("side_branches=%d batches_in_flight=%d: %.0f objects/s  (%.3f ms per step)" % (branch, streams, d["value"], d["ms_per_step"]), flush=True)
        except Exception as e:
            print("side_branches=%d batches_in_flight=%d: FAILED %s" % (branch, streams, e), flush=True)
