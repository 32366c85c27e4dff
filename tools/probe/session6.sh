#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
echo "== pipeline tests"; timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_js_e2e_gpu.py -x -q > gpurun_out/s6/tests.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/s6/tests.log
echo "== pipeline bench"; timeout -k 10 300 python tools/pipeline_bench.py --loop 480 > gpurun_out/s6/pipe.json 2> gpurun_out/s6/pipe.err; echo "rc=$?"; cat gpurun_out/s6/pipe.json; tail -c 300 gpurun_out/s6/pipe.err
echo "== node pipeline bench"; timeout -k 10 300 node tools/js_pipeline_bench.js tools/probe/stream_1080p_2gop.bin --loop 480 --threads 16 --window 32 > gpurun_out/s6/pipe_node.json 2> gpurun_out/s6/pipe_node.err; echo "rc=$?"; cat gpurun_out/s6/pipe_node.json; tail -c 300 gpurun_out/s6/pipe_node.err
echo done
echo "== thread scaling (informational)"; for t in 4 8 32 64; do timeout -k 10 120 python tools/pipeline_bench.py --loop 240 --threads $t 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($t, 'threads:', round(d['value']), 'pictures/s; per-thread parse', round(d['parser_pictures_per_s_per_thread']))"; done
