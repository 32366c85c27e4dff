#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE
# in their own --pmc runs (MI355X_MICROARCH.md: never combined with tracing).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "trace";  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/kt.log 2>&1 || echo "trace failed"
echo "fetch";  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pf --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pf.log 2>&1 || echo "fetch failed"
echo "write";  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pw --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pw.log 2>&1 || echo "write failed"
echo done
