#!/bin/bash
# L2 / fabric side of the GPU parser's kernels beside the reconstruction (pipeline bench, gpu_parser): how many requests,
# how many of the writes are whole 64-byte lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/tcc_vlc}
mkdir -p $out
S="python3 tools/pipeline_bench.py --varied --loop 96 --threads 16 --window 128 --inflight 3 --gpu-parser"
i=0
for c in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_ATOMIC_sum TCC_REQ_sum" "TCC_WRITE_sum TCC_READ_sum TCC_MISS_sum" "TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_HIT_sum"; do
  i=$((i+1)); timeout -k 10 200 rocprofv3 --pmc $c -d $out/p$i --output-format csv -- $S > $out/p$i.log 2>&1 || echo "pass $i ($c) failed"
done
python3 - $out <<'PY' > $out.txt
import csv,glob,collections,sys
pmc=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "leon::k_" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]),float(r["Counter_Value"])))
    for k,cs in agg.items():
        for c,vals in cs.items():
            g=max(x for x,_ in vals); big=[v for x,v in vals if x==g]
            pmc[k][c]=sum(big)/len(big)
for k in sorted(pmc):
    print(k)
    print('   ', {c:"%.3g" % v for c,v in sorted(pmc[k].items())})
PY
tail -3 $out/p*.log | cut -c1-300 >> $out.txt
rm -rf $out
cat $out.txt
