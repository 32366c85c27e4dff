#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
timeout -k 10 600 python -m pytest tests/test_alpha_gpu.py tests/test_parity_gpu.py -x -q > gpurun_out/s8/tests.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/s8/tests.log
