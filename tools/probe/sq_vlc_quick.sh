#!/bin/bash
# Two --pmc passes over the GPU parser's kernels (lane utilisation and instruction counts): bash tools/probe/sq_vlc_quick.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/sq_vlc_q}
mkdir -p $out
S="python3 tools/pipeline_bench.py --varied --loop 96 --threads 16 --window 128 --inflight 3 --gpu-parser"
i=0
for c in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1)); timeout -k 10 200 rocprofv3 --pmc $c -d $out/p$i --output-format csv -- $S > $out/p$i.log 2>&1 || echo "pass $i ($c) failed"
done
python3 - $out <<'PY' > $out.txt
import csv,glob,collections,sys
pmc=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "leon::k_vlc" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]),float(r["Counter_Value"])))
    for k,cs in agg.items():
        for c,vals in cs.items():
            g=max(x for x,_ in vals); big=[v for x,v in vals if x==g]
            pmc[k][c]=sum(big)/len(big)
for k in sorted(pmc):
    w=pmc[k].get("SQ_WAVES",1)
    print(k, "waves", w)
    print('   ', {c:round(v/w,1) for c,v in sorted(pmc[k].items())})
    if "SQ_THREAD_CYCLES_VALU" in pmc[k] and pmc[k].get("SQ_ACTIVE_INST_VALU"):
        print('    active lanes per vector instruction: %.1f of 64' % (pmc[k]["SQ_THREAD_CYCLES_VALU"] / pmc[k]["SQ_ACTIVE_INST_VALU"]))
PY
rm -rf $out
cat $out.txt
