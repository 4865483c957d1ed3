"""Development (GPU box): A/B of two builds of the library on the same box, alternating, eval forward at B=32 (graph replay, two
batches in flight, as bench.py's default):  python scripts/ab_bench.py libtgpose_hip.so libtgpose_hip_noguard.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
code = """
import sys, os, json
sys.path.insert(0, %r)
from tgpose_amd import _lib
_lib.LIB_PATH = os.path.join(%r, 'tg-pose_amd', sys.argv[1])
sys.argv = ['bench.py', '--no-cpu-baseline'] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(%r, 'bench.py'), run_name='__main__')
""" % (ROOT, ROOT, ROOT)
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code, l] + sys.argv[4:], capture_output=True, text=True).stdout.strip().splitlines()
        d = json.loads(out[-1])
        res[l].append((d["value"], d["roofline"]["avg_launch_us"], d["config"].get("objects_per_s_one_batch_in_flight")))
        print(l, res[l][-1], flush=True)
for l in libs:
    v = sorted(x[0] for x in res[l])
    print("%-32s median %.0f objects/s  (%s)" % (l, v[len(v) // 2], ", ".join("%.0f" % x for x in v)))
