#!/usr/bin/env python3
"""Condense the rocprofv3 passes over tools/probe/fetch_calib.bin into profiles/<tag>_fetch_calibration.json.

  python tools/summarize_calibration.py <tag> <plain.json> <dir with --pmc FETCH_SIZE> <dir with raw TCC counters>

Per load shape: the known unique byte count, FETCH_SIZE (KB as reported), the raw TCC_EA0 read-request
counters, and the factors  unique_bytes / (FETCH_SIZE * 1024)  and  bytes per read request."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d):
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                out[k][c] = v[-1]          # the probe runs every kernel twice; the second pass is the warm one
    return out


def main():
    tag, plain, dfetch, draw = sys.argv[1:5]
    known = json.load(open(plain))["kernels"]
    cf, cr = counters(dfetch), counters(draw)
    res = {}
    for k, e in known.items():
        fs = cf.get(k, {}).get("FETCH_SIZE")
        rd = cr.get(k, {}).get("TCC_EA0_RDREQ_sum")
        res[k] = {"unique_bytes": e["unique_bytes"], "gbps_unprofiled": e["gbps"], "FETCH_SIZE_KB": fs,
                  "TCC_EA0_RDREQ_sum": rd, "TCC_EA0_RDREQ_32B_sum": cr.get(k, {}).get("TCC_EA0_RDREQ_32B_sum"),
                  "TCC_BUBBLE_sum": cr.get(k, {}).get("TCC_BUBBLE_sum"),
                  "unique_bytes_per_FETCH_SIZE_byte": e["unique_bytes"] / (fs * 1024) if fs else None,
                  "unique_bytes_per_read_request": e["unique_bytes"] / rd if rd else None,
                  "FETCH_SIZE_bytes_per_read_request": fs * 1024 / rd if (fs and rd) else None}
    note = ("gfx950, rocprofv3 (ROCm 7.2): FETCH_SIZE = TCC_EA0_RDREQ_sum * 64 B (TCC_BUBBLE and RDREQ_32B read 0), while every "
            "read request of these shapes moves one whole 128-byte line -- also when only 4 bytes of each 64-byte half are "
            "used (calib_b32_half).  So HBM/fabric read bytes = 2 * FETCH_SIZE * 1024 = 128 * TCC_EA0_RDREQ_sum for every "
            "load shape k_recon issues; what differs per shape is how many lines are requested (calib_b96_win: lines shared "
            "by workgroups on different XCDs are requested once per XCD L2).")
    json.dump({"note": note, "shapes": res}, open(os.path.join(ROOT, "profiles", tag + "_fetch_calibration.json"), "w"), indent=1)
    for k, v in res.items():
        print(k, v["unique_bytes_per_FETCH_SIZE_byte"], v["unique_bytes_per_read_request"])


if __name__ == "__main__":
    main()
