#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then counters in their own --pmc runs, never
# combined with tracing (MI355X_MICROARCH.md).  Usage on the GPU box:  bash tools/probe/profile_round.sh <outdir>
# then, back here:  python tools/summarize_profile.py <tag> --stats <outdir>/kt --pmc <outdir>/pf <outdir>/pw <outdir>/ptcp <outdir>/valu
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/prof}
mkdir -p $out
# (--no-placement: the start-up step that draws every region twice launches the same kernels on the draws it rejects; the trace
#  and the counters are to show the launches of the steps)
B="python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --no-placement"
S="$B --steps 2 --warmup 1"
echo "== trace";  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- $B > $out/kt.log 2>&1 || echo "trace failed"
echo "== fetch";  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $out/pf --output-format csv -- $S > $out/pf.log 2>&1 || echo "fetch failed"
echo "== write";  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $out/pw --output-format csv -- $S > $out/pw.log 2>&1 || echo "write failed"
echo "== tcp";    timeout -k 10 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $out/ptcp --output-format csv -- $S > $out/ptcp.log 2>&1 || echo "tcp failed"
echo "== valu";   timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $out/valu --output-format csv -- $S > $out/valu.log 2>&1 || echo "valu failed"
# the memory pipe, 2-3 counters per pass (more "exceeds the capabilities of the hardware to collect")
i=0
[ -n "$LEON_MEMPIPE" ] && for c in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WRITE_WAVEFRONTS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1)); timeout -k 10 400 rocprofv3 --pmc $c -d $out/mp$i --output-format csv -- $S --unique 2 > $out/mp$i.log 2>&1 || echo "memory-pipe pass $i failed"
done
# condense on the box (the raw CSVs of a round exceed what gpurun copies back) and drop the raw output
tag=${2:-round}
python3 tools/summarize_profile.py $tag --out $out/summary --stats $out/kt --pmc $out/pf $out/pw $out/ptcp $out/valu \
    --note "bench.py default workload: 128 GOPs, 8 GOP bodies, random motion, display conversion fused into the reconstruction kernels; read bytes = 2 x FETCH_SIZE x 1024 = 128 B per TCC_EA0_RDREQ, calibrated in profiles/r02_fetch_calibration.json"
grep -h "^{" $out/kt.log | tail -1 > $out/summary/${tag}_bench_line_under_rocprofv3.json
rm -rf $out/kt $out/pf $out/pw $out/ptcp $out/valu $out/mp1 $out/mp2 $out/mp3 $out/mp4
# the calibration of FETCH_SIZE for this kernel's load shapes (tools/summarize_calibration.py)
if [ -n "$LEON_CALIBRATE" ]; then
echo "== calibration"; ./tools/probe/fetch_calib.bin > $out/calib_plain.json
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $out/cal_fetch --output-format csv -- ./tools/probe/fetch_calib.bin > $out/cal_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d $out/cal_raw --output-format csv -- ./tools/probe/fetch_calib.bin > $out/cal_raw.log 2>&1
fi
timeout -k 10 400 python bench.py > $out/summary/${tag}_bench_line.json 2> $out/bench.err; echo "bench rc=$?"
echo done
