"""Development (GPU box): more batches in flight, branch-free captured forwards."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for branch, streams in ((1, 2), (0, 4), (0, 5), (0, 6), (0, 8), (0, 12), (1, 2), (0, 4), (0, 6)):
    args = ["--no-cpu-baseline", "--streams", str(streams), "--min-seconds", "0.7"] + ([] if branch else ["--no-branch-streams"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True).stdout.strip().splitlines()
    try:
        d = json.loads(out[-1])
        print("side_branches=%d batches_in_flight=%d: %.0f objects/s  (%.3f ms per step)" % (branch, streams, d["value"], d["ms_per_step"]), flush=True)
    except Exception as e:
        print("side_branches=%d batches_in_flight=%d: FAILED %s" % (branch, streams, e), flush=True)
