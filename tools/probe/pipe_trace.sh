#!/bin/bash
# Kernel timeline of the pipeline with the GPU parser: rocprofv3 --kernel-trace of tools/pipeline_bench.py, condensed to
# per-kernel sums, the time the GPU ran anything at all, and the time per window.  Usage: bash tools/probe/pipe_trace.sh <outdir> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/kt --output-format csv -- python3 tools/pipeline_bench.py --gpu-parser --threads 16 --window 128 --inflight 3 "$@" > $out/bench.json 2> $out/err.log || echo "trace failed"
python3 - $out <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/kt/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void leon::", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
per = collections.defaultdict(lambda: [0, 0])
for a, b, k in rows:
    per[k][0] += 1; per[k][1] += b - a
busy = 0; cur_a, cur_b = rows[0][0], rows[0][1]
for a, b, _ in rows[1:]:
    if a > cur_b: busy += cur_b - cur_a; cur_a, cur_b = a, b
    else: cur_b = max(cur_b, b)
busy += cur_b - cur_a
# idle gaps of the whole device: how long, and which kernel ended them
gaps = collections.defaultdict(lambda: [0, 0])
end = rows[0][1]
for a, b, k in rows[1:]:
    if a > end:
        gaps[k][0] += 1; gaps[k][1] += a - end
    end = max(end, b)
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
n_parse = max(1, max([v[0] for k, v in per.items() if "k_vlc_parse" in k] + [1]))
res = {"pictures_per_s": bench["value"], "windows": n_parse, "span_ms": (t1 - t0) / 1e6, "gpu_busy_ms": busy / 1e6, "ms_per_window": (t1 - t0) / 1e6 / max(1, n_parse),
       "idle_before": {k: {"gaps": v[0], "total_ms": round(v[1] / 1e6, 2)} for k, v in sorted(gaps.items(), key=lambda x: -x[1][1])},
       "kernels": {k: {"launches": v[0], "total_ms": round(v[1] / 1e6, 2), "avg_ms": round(v[1] / 1e6 / v[0], 3), "ms_per_window": round(v[1] / 1e6 / max(1, n_parse), 3)} for k, v in sorted(per.items(), key=lambda x: -x[1][1])}}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $out/kt
