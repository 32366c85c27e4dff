#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<dir>/...) into small files under profiles/.

  python tools/summarize_profile.py <round-tag> --stats <dir> [--pmc <dir> ...]

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table) and
profiles/<tag>_pmc.json (per kernel: mean counter value over the dispatches with the
largest grid, FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for
wide coalesced reads on gfx950; WRITE_SIZE exact; both in bytes)."""
import argparse, collections, csv, glob, json, os, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--note", default="")
    ap.add_argument("--timed-launches", type=int, default=0,
                    help="also write <tag>_kernel_stats_timed_region.csv: rocprofv3's stats columns over the LAST n reconstruction dispatches of the "
                         "trace only (bench.py's timed steps x 8 launches, when nothing behind them launches k_recon*: --no-cpu-baseline "
                         "--no-second-recipe --no-end-to-end) -- so that a run WITH the start-up placement step, whose draws launch the same kernels, can be traced")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"), help="directory to write to (on the GPU box: somewhere under gpurun_out/)")
    a = ap.parse_args()
    out = a.out
    os.makedirs(out, exist_ok=True)
    if a.stats:
        f = glob.glob(os.path.join(a.stats, "**", "*_kernel_stats.csv"), recursive=True)[0]
        shutil.copy(f, os.path.join(out, a.tag + "_kernel_stats.csv"))
        tr = glob.glob(os.path.join(a.stats, "**", "*_kernel_trace.csv"), recursive=True)
        if tr:   # per-dispatch durations of our kernels only
            rows = [r for r in csv.DictReader(open(tr[0])) if "leon::" in r["Kernel_Name"]]
            with open(os.path.join(out, a.tag + "_kernel_trace_leon.csv"), "w") as w:
                w.write("kernel,grid_x,duration_us\n")
                for r in rows:
                    w.write('"%s",%s,%.3f\n' % (r["Kernel_Name"].split("(")[0], r["Grid_Size_X"],
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            if a.timed_launches > 0:
                rec = sorted((r for r in rows if "k_recon" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))[-a.timed_launches:]
                by = collections.OrderedDict()
                for r in rec:
                    by.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                tot = sum(sum(v) for v in by.values())
                with open(os.path.join(out, a.tag + "_kernel_stats_timed_region.csv"), "w") as w:
                    w.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
                    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                        w.write('"%s",%d,%d,%.6f,%.2f,%d,%d\n' % (k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
    pmc = collections.defaultdict(dict)
    for d in a.pmc:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                if "leon::" not in r["Kernel_Name"]:
                    continue
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            for k, cs in agg.items():
                for c, vals in cs.items():
                    g = max(x for x, _ in vals)
                    big = [v for x, v in vals if x == g]
                    pmc[k][c] = {"grid_size": g, "dispatches": len(big), "mean": sum(big) / len(big)}
    for k, cs in pmc.items():
        if "FETCH_SIZE" in cs:
            cs["hbm_read_bytes_corrected"] = cs["FETCH_SIZE"]["mean"] * 1024 * 2
        if "WRITE_SIZE" in cs:
            cs["hbm_write_bytes"] = cs["WRITE_SIZE"]["mean"] * 1024
    if pmc:
        json.dump({"note": a.note, "kernels": pmc}, open(os.path.join(out, a.tag + "_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
