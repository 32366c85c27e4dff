#!/bin/bash
# Vector / scalar / LDS instructions per wave of the reconstruction kernels for several library variants (tools/ab_build.sh
# names), one --pmc pass each, no tracing.  Usage on the GPU box: bash tools/probe/valu_variants.sh <outdir> <variant>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $out
S="python3 bench.py --no-cpu-baseline --no-second-recipe --steps 2 --warmup 1 --unique 2"
for v in "$@"; do
  if [ "$v" != "tree" ]; then export LEON_DEBUG_LIB=$GRAFT_REPO_ROOT/build/ab/$v/libleon_hip.so; else unset LEON_DEBUG_LIB; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU -d $out/$v --output-format csv -- $S > $out/$v.log 2>&1 || echo "$v failed"
  python3 tools/probe/sq_summary.py $out/$v >> $out/summary.txt && rm -rf $out/$v
done
cat $out/summary.txt
