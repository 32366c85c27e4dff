#!/bin/bash
# usage: pmc_passes.sh <outdir-prefix> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pfx=$1; shift
i=0
for c in "$@"; do
  i=$((i+1))
  echo "pass $i: $c"; timeout -k 10 200 rocprofv3 --pmc $c -d gpurun_out/${pfx}$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rgba --gops 48 > gpurun_out/${pfx}$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${pfx}$i.log; }
done
echo done
