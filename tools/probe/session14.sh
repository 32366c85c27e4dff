#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s14
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q > gpurun_out/s14/tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/s14/tests.log
