"""Development (GPU box): the eval forward's optional side branches on / off, one and two batches in flight."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for coarse, tail in ((0, 0), (1, 0), (0, 1), (1, 1)):
    for streams in (1, 2):
        env = dict(os.environ, TGP_COARSE_SIDE=str(coarse), TGP_HEADS_TAIL=str(tail))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--streams", str(streams)] + sys.argv[1:],
                             capture_output=True, text=True, env=env).stdout.strip().splitlines()
        d = json.loads(out[-1])
        print("coarse_side=%d heads_tail=%d streams=%d: %.0f objects/s  (%.3f ms)" % (coarse, tail, streams, d["value"], d["ms_per_step"]), flush=True)
