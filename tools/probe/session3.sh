#!/bin/bash
# rocprofv3 passes behind profiles/r02_*: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and L1 counters in their own --pmc runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s11
echo "== trace";  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s11/kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe > gpurun_out/s11/kt.log 2>&1 || echo "trace failed"
echo "== fetch";  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/s11/pf --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s11/pf.log 2>&1 || echo "fetch failed"
echo "== write";  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/s11/pw --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s11/pw.log 2>&1 || echo "write failed"
echo "== tcp";    timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/s11/ptcp --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s11/ptcp.log 2>&1 || echo "tcp failed"
echo "== valu";   timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d gpurun_out/s11/valu --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe > gpurun_out/s11/valu.log 2>&1 || echo "valu failed"
timeout -k 10 400 python bench.py > gpurun_out/s11/bench.json 2> gpurun_out/s11/bench.err; echo bench rc=$?
tail -n 2 gpurun_out/s11/kt.log | cut -c1-300
echo done
