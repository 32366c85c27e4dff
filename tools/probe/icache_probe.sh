#!/bin/bash
# Instruction-cache requests / misses of the kernels, alone (bench.py) and side by side (the pipeline with the GPU parser)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; mkdir -p $out
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
timeout -k 10 200 rocprofv3 --pmc $C -d $out/bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe --steps 2 --warmup 1 --unique 2 --boundary sparse > $out/bench.log 2>&1 || echo "bench pass failed"
timeout -k 10 200 rocprofv3 --pmc $C -d $out/pipe --output-format csv -- python3 tools/pipeline_bench.py --gpu-parser --threads 16 --window 128 --inflight 3 --varied --loop 96 > $out/pipe.log 2>&1 || echo "pipe pass failed"
python3 - $out <<'PY'
import csv, glob, sys, collections
for which in ("bench", "pipe"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(sys.argv[1] + "/" + which + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void leon::", "").replace("leon::", "")
            if "k_recon" in k or "k_vlc" in k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
    print(which)
    for k, c in sorted(agg.items()):
        req = c.get("SQC_ICACHE_REQ", 0)
        print("  %-44s dispatches %4d  requests %.3g  misses %.3g  miss rate %.4f" % (k, n[k], req, c.get("SQC_ICACHE_MISSES", 0), c.get("SQC_ICACHE_MISSES", 0) / req if req else 0))
PY
rm -rf $out/bench $out/pipe
