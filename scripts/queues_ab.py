"""Development (GPU box): hipGraph's shared pool of parallel streams (DEBUG_HIP_FORCE_GRAPH_QUEUES) and the hardware queue count
(GPU_MAX_HW_QUEUES) against the forward's side branches, one and two batches in flight."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for gq, hq in ((None, None), (8, None), (8, 8), (16, 8), (None, 8)):
    for coarse, tail in ((0, 0), (1, 1), (1, 0)):
        for streams in (1, 2):
            env = dict(os.environ, TGP_COARSE_SIDE=str(coarse), TGP_HEADS_TAIL=str(tail))
            if gq:
                env["DEBUG_HIP_FORCE_GRAPH_QUEUES"] = str(gq)
            if hq:
                env["GPU_MAX_HW_QUEUES"] = str(hq)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--streams", str(streams), "--min-seconds", "0.5"],
                                 capture_output=True, text=True, env=env).stdout.strip().splitlines()
            try:
                d = json.loads(out[-1])
                print("graph_queues=%s hw_queues=%s coarse_side=%d heads_tail=%d streams=%d: %.0f objects/s  (%.3f ms)"
                      % (gq, hq, coarse, tail, streams, d["value"], d["ms_per_step"]), flush=True)
            except Exception as e:
                print("graph_queues=%s hw_queues=%s coarse=%d tail=%d streams=%d: FAILED %s" % (gq, hq, coarse, tail, streams, e), flush=True)
