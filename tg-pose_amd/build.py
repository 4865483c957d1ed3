"""Builds tg-pose_amd/libtgpose_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m tgpose_amd.build          # or: from tgpose_amd.build import build; build()

Flags that matter:
  --offload-arch=gfx950   the only target (CDNA4 / MI355X)
  -ffp-contract=off       the kNN / Chamfer kernels mirror the CPU path's fp32 rounding exactly;
                          fused multiply-adds appear only where written (fmaf, MFMA)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["knn.hip", "gemm.hip", "gconv.hip", "chamfer.hip", "dcd.hip", "bn.hip", "backward.hip", "gconv_bwd.hip", "evalmetrics.hip", "tdaloss.hip", "inputside.hip", "rowsort.hip", "version.hip", "gemm_variants.hip"]
LIB = os.path.join(HERE, "libtgpose_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-variable", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "tgpose.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
