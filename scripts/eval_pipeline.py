"""GPU box: end-to-end rate of the evaluation driver on synthetic frames held in host memory (BASELINE config 5 without the
dataset, which is not available offline): depth frame + detection masks -> clouds -> PoseNet9D.forward -> pose assembly ->
pred_results, through tgpose_amd.evaluater.RT_TDA_Evaluater.myEvaluater.  Host -> device copies of the frames are INSIDE the
timed region (78 MB per 32 frames).  Prints one JSON line per sampler.

    python scripts/eval_pipeline.py [frames=128] [detections per frame=6]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tests.util import synth_depth_scene
from tgpose_amd import PoseNet9D, seeded_state_dict, FLAGS
from tgpose_amd.evaluater.RT_TDA_Evaluater import myEvaluater

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
net = PoseNet9D()
net.load_state_dict(seeded_state_dict(0), strict=True)
net = net.to(dev).eval()
FLAGS.train = 0
records = [dict(frame=synth_depth_scene(7000 + i, per)) for i in range(n_frames)]
for sampler, graph in (("numpy", False), ("device", False), ("device", True)):
    ev = myEvaluater(net, frames_per_batch=32, max_batch=192, sampler=sampler, seed=1, graph=graph)
    np.random.seed(0)
    torch.manual_seed(0)
    ev.run(records[:32])                                     # warm-up: allocator, first launches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ev.run(records)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    objs = sum(len(r["pred_RTs"]) for r in res)
    print(json.dumps({"metric": "evaluation driver end to end (host frames -> pred_results)", "sampler": sampler, "hipgraph": graph, "frames": len(res),
                      "objects": objs, "seconds": round(dt, 4), "frames_per_s": round(len(res) / dt, 1), "objects_per_s": round(objs / dt, 1),
                      "frames_per_batch": 32, "max_batch": 192,
                      "note": "PCIe upload of depth + masks, cloud building, forwards of up to 192 objects, pose assembly and the "
                              "device-to-host copy of the results are all inside the timed region; a chunk's results are fetched "
                              "after the next chunk has been enqueued"}), flush=True)
