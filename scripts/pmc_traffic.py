"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into the per-kernel
traffic table bench.py reads (profiles/r01_*_pmc_traffic.json).

    python scripts/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE reports half of a coalesced 16-B-per-lane
read stream (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0][:120]].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        out[k] = dict(launches=len(fetch.get(k, write.get(k, []))), fetch_kb_per_launch=round(f, 1),
                      write_kb_per_launch=round(w, 1), bytes_per_launch_corrected=int((2 * f + w) * 1024))
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 2 "
            "--no-cpu-baseline` (6 forwards). Values are KB per launch as rocprofv3 reports them; bytes = "
            "(2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half of a coalesced 16-B-per-lane read stream "
            "(MI355X_MICROARCH.md, HBM section). FETCH_SIZE counts L2 misses to the fabric, Infinity-Cache hits included.")
    json.dump(dict(note=note, kernels=out), open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if "gemm" in k:
            print(k, v)


if __name__ == "__main__":
    main()
