#!/bin/bash
# vector instructions per wave of the reconstruction kernels (one --pmc pass of bench.py): bash tools/probe/valu_pass.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/valu}; mkdir -p $out
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/raw --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --no-placement --steps 2 --warmup 1 > $out/log.txt 2>&1 || echo "pass failed"
python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/raw/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "leon::k_recon" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, cs in sorted(agg.items()):
    m = {}
    for c, vals in cs.items():
        g = max(x for x, _ in vals); big = [v for x, v in vals if x == g]; m[c] = sum(big) / len(big)
    print(k, "VALU per wave %.1f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"]), "waves %d" % m["SQ_WAVES"])
PY
rm -rf $out/raw
