#!/bin/bash
# round-2 GPU session 2: calibration (fixed byte-load shape), profile passes of the default bench, 2-rank rehearsal
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
echo "== calib plain"; timeout -k 10 120 ./tools/probe/fetch_calib.bin > gpurun_out/s2/calib_plain.json 2>&1; cat gpurun_out/s2/calib_plain.json
echo "== calib FETCH_SIZE"; timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/s2/cal_fetch --output-format csv -- ./tools/probe/fetch_calib.bin > gpurun_out/s2/cal_fetch.log 2>&1; echo rc=$?
echo "== calib raw"; timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d gpurun_out/s2/cal_raw --output-format csv -- ./tools/probe/fetch_calib.bin > gpurun_out/s2/cal_raw.log 2>&1; echo rc=$?
echo "== trace";  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/s2/kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe > gpurun_out/s2/kt.log 2>&1 || echo "trace failed"
echo "== fetch";  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/s2/pf --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s2/pf.log 2>&1 || echo "fetch failed"
echo "== write";  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/s2/pw --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s2/pw.log 2>&1 || echo "write failed"
echo "== tcp";    timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/s2/ptcp --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s2/ptcp.log 2>&1 || echo "tcp failed"
echo "== 2 ranks on one device over gloo (rehearsal of --gpus 2)"
LEON_BENCH_BACKEND=gloo LEON_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 --gops 64 --steps 10 --warmup 3 > gpurun_out/s2/two_ranks.json 2> gpurun_out/s2/two_ranks.err; echo rc=$?; tail -c 400 gpurun_out/s2/two_ranks.err
echo done
