#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s13
timeout -k 10 300 python tools/pipeline_bench.py --loop 480 > gpurun_out/s13/pipe.json 2> gpurun_out/s13/pipe.err; echo "rc=$?"; cat gpurun_out/s13/pipe.json
timeout -k 10 300 node tools/js_pipeline_bench.js tools/probe/stream_1080p_2gop.bin --loop 480 --threads 16 --window 32 > gpurun_out/s13/pipe_node.json 2> gpurun_out/s13/pipe_node.err; echo "rc=$?"; cat gpurun_out/s13/pipe_node.json
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_sparse_gpu.py tests/test_glsl_golden_gpu.py -x -q > gpurun_out/s13/tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/s13/tests.log
