#!/bin/bash
# the counter passes of tools/probe/profile_round.sh on their own (no trace, no bench line): profiles/<tag>_pmc.json
#   TAG=r04f bash tools/probe/pmc_round.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc}; tag=${TAG:-round}
mkdir -p $out
S="python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --no-placement --steps 2 --warmup 1"
echo "== fetch";  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/pf --output-format csv -- $S > $out/pf.log 2>&1 || echo "fetch failed"
echo "== write";  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/pw --output-format csv -- $S > $out/pw.log 2>&1 || echo "write failed"
echo "== tcp";    timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $out/ptcp --output-format csv -- $S > $out/ptcp.log 2>&1 || echo "tcp failed"
echo "== valu";   timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $out/valu --output-format csv -- $S > $out/valu.log 2>&1 || echo "valu failed"
python3 tools/summarize_profile.py $tag --out $out/summary --pmc $out/pf $out/pw $out/ptcp $out/valu \
    --note "bench.py default workload: 128 GOPs, 8 GOP bodies, random motion, display conversion fused into the reconstruction kernels; read bytes = 2 x FETCH_SIZE x 1024 = 128 B per TCC_EA0_RDREQ, calibrated in profiles/r02_fetch_calibration.json"
rm -rf $out/pf $out/pw $out/ptcp $out/valu
ls $out/summary
