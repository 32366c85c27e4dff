#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
echo "== fused tests"; timeout -k 10 600 python -m pytest tests/test_fused_display_gpu.py tests/test_parity_gpu.py tests/test_sparse_gpu.py -x -q > gpurun_out/s4/tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/s4/tests.log
echo "== bench fused"; timeout -k 10 400 python bench.py --no-second-recipe > gpurun_out/s4/bench_fused.json 2> gpurun_out/s4/bench_fused.err; echo "rc=$?"; tail -c 300 gpurun_out/s4/bench_fused.err
echo "== bench unfused"; timeout -k 10 400 python bench.py --no-second-recipe --no-fuse --no-cpu-baseline > gpurun_out/s4/bench_unfused.json 2> gpurun_out/s4/bench_unfused.err; echo "rc=$?"
echo done
