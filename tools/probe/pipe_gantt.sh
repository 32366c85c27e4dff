#!/bin/bash
# A text Gantt chart of a few steady-state windows of the pipeline (kernel trace of tools/pipeline_bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/kt --output-format csv -- python3 tools/pipeline_bench.py --gpu-parser --threads 16 --inflight 3 "$@" > $out/bench.json 2> $out/err.log || echo "trace failed"
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/kt/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void leon::", "").replace("leon::", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
parses = [i for i, r in enumerate(rows) if r[2] == "k_vlc_parse"]
if len(parses) > 70:
    i0, i1 = parses[60], parses[64]
    t0 = rows[i0][0]
    with open(out + "/gantt.txt", "w") as f:
        for a, b, k, q, s in rows:
            if a >= t0 - 2_000_000 and a <= rows[i1][1]:
                f.write("%9.3f ms  +%7.3f ms  q%-3s %s\n" % ((a - t0) / 1e6, (b - a) / 1e6, q, k))
    print(open(out + "/gantt.txt").read()[:9000])
PY
rm -rf $out/kt
