#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
echo "== all gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s7/tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/s7/tests.log
echo done
