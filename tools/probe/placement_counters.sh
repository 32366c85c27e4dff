#!/bin/bash
# Which counters follow the placement of the RGBA frames?  tools/probe/spread_probe.py --hold 8 (one I launch against eight
# live RGBA buffers x three sets of coefficient planes) under rocprofv3 --pmc, counters per dispatch, beside the launch
# times the probe measures itself in the same process.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/placement}; mkdir -p $out
i=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $c -d $out/p$i --output-format csv -- python3 tools/probe/spread_probe.py --hold 8 --launches 4 > $out/p$i.json 2> $out/p$i.err || echo "pass $i failed"
  python3 - $out/p$i $out/p$i.json <<'PY'
import csv, glob, json, sys, collections
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_recon_display" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
except Exception as e:
    print("no probe output", e); sys.exit(0)
ms = j["rgba_x_coef_median_ms"]
per = len(ids) // (len(ms) * len(ms[0])) if ms else 0
print("dispatches", len(ids), "per cell", per)
names = sorted({k for r in rows.values() for k in r})
for b in range(len(ms)):
    cells = []
    for s in range(len(ms[0])):
        chunk = ids[(b * len(ms[0]) + s) * per:(b * len(ms[0]) + s + 1) * per][-4:]
        cells.append({n: sum(rows[i].get(n, 0) for i in chunk) / max(1, len(chunk)) for n in names})
    avg = {n: sum(c[n] for c in cells) / len(cells) for n in names}
    print("buffer %d  %.4f ms   " % (b, sum(ms[b]) / len(ms[b])) + "  ".join("%s %.4g" % (n.replace("_sum", ""), v) for n, v in avg.items()))
PY
  rm -rf $out/p$i
done
