#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_js_e2e_gpu.py -x -q > gpurun_out/s10/tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/s10/tests.log
echo "== two Node shards on one device"; timeout -k 10 200 node tools/js_multi_gpu.js tools/probe/stream_1080p_2gop.bin --gpus 2 --one-device --loop 200 --threads 8 --window 32 > gpurun_out/s10/js2.json 2> gpurun_out/s10/js2.err; echo rc=$?; cat gpurun_out/s10/js2.json; tail -c 300 gpurun_out/s10/js2.err
