"""Turns rocprofv3 --pmc passes (separate runs, kernel-trace only) over serial eager forwards into the per-kernel SQ table of
profiles/r*_sq_counters_forward.txt:   python scripts/sq_counters.py <out.txt> <pass dir> [<pass dir> ...]
parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier), stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, issuing = SQ_ACTIVE_INST_ANY /
SQ_WAVE_CYCLES, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), LDS conflict = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE (cycles an LDS instruction spent in bank conflicts over the cycles the LDS was busy)."""
import collections, csv, glob, os, sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(int)
    dur = collections.defaultdict(float)
    seen_calls = set()
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (d, r["Dispatch_Id"])
                if key not in seen_calls and d == dirs[0]:
                    seen_calls.add(key)
                    calls[k] += 1
                    if "Start_Timestamp" in r and "End_Timestamp" in r:
                        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = []
    for k, c in acc.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0 or "at::native" in k or "rocclr" in k:
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mfma = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (gui / 8.0) if gui else float("nan")
        lds = 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        rows.append((dur[k], k, calls[k], dur[k] / max(calls[k], 1) / 1e3, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
                     100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, mfma, lds, c.get("SQ_INSTS_MFMA", 0) / max(calls[k], 1), c.get("SQ_INSTS_VALU", 0) / max(calls[k], 1)))
    rows.sort(reverse=True)
    with open(out, "w") as fo:
        fo.write(__doc__.strip() + "\n\n")
        fo.write("%-52s %6s %9s %8s %8s %9s %10s %13s %12s %12s\n" % ("kernel", "calls", "avg us", "parked%", "stalled%", "issuing%", "MFMA busy%",
                                                                  "LDS conflict%", "MFMA insts", "VALU insts"))
        for _, k, n, us, pk, stl, iss, mf, lds, im, iv in rows:
            fo.write("%-52s %6d %9.1f %8.0f %8.0f %9.0f %10.0f %13.0f %12.3g %12.3g\n" % (k, n, us, pk, stl, iss, mf, lds, im, iv))
    print(open(out).read())


if __name__ == "__main__":
    main()
