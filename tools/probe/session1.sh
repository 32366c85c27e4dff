#!/bin/bash
# round-2 GPU session 1: tests, bench sanity, counter list, FETCH_SIZE calibration, VALU counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
echo "== tests"; timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s1/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s1/tests.log
echo "== bench"; timeout -k 10 400 python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/s1/bench.err
echo "== counters"; timeout -k 10 60 rocprofv3 -L > gpurun_out/s1/counters.txt 2>&1; grep -c . gpurun_out/s1/counters.txt
echo "== calib plain"; timeout -k 10 120 ./tools/probe/fetch_calib.bin > gpurun_out/s1/calib_plain.json 2>&1; cat gpurun_out/s1/calib_plain.json
echo "== calib FETCH_SIZE"; timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/s1/cal_fetch --output-format csv -- ./tools/probe/fetch_calib.bin > gpurun_out/s1/cal_fetch.log 2>&1; echo rc=$?
echo "== calib raw"; timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d gpurun_out/s1/cal_raw --output-format csv -- ./tools/probe/fetch_calib.bin > gpurun_out/s1/cal_raw.log 2>&1; echo rc=$?; tail -3 gpurun_out/s1/cal_raw.log
echo "== valu"; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d gpurun_out/s1/valu --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe --unique 2 > gpurun_out/s1/valu.log 2>&1; echo rc=$?
echo done
