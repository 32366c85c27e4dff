#!/bin/bash
# Issue-side counters of the reconstruction kernels for one library variant (tools/ab_build.sh), 3 counters per
# --pmc pass, no tracing.  Usage on the GPU box: bash tools/probe/sq_passes.sh <variant|tree> <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
v=$1; out=${2:-gpurun_out/sq_$v}
mkdir -p $out
if [ "$v" != "tree" ]; then export LEON_DEBUG_LIB=$GRAFT_REPO_ROOT/build/ab/$v/libleon_hip.so; fi
S="python3 bench.py --no-cpu-baseline --no-second-recipe --steps 2 --warmup 1 --unique 2"
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); timeout -k 10 200 rocprofv3 --pmc $c -d $out/p$i --output-format csv -- $S > $out/p$i.log 2>&1 || echo "pass $i ($c) failed"
done
python3 tools/probe/sq_summary.py $out > $out.txt && rm -rf $out
echo done
