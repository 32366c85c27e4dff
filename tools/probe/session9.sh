#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s9/tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/s9/tests.log
