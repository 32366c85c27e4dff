#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s15
timeout -k 10 600 python -m pytest tests/test_bench_multi_rank_gpu.py -x -q > gpurun_out/s15/tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/s15/tests.log
timeout -k 10 400 python bench.py --boundary sparse --no-second-recipe --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/s15/sparse.json 2> gpurun_out/s15/sparse.err; echo "sparse rc=$?"; tail -c 300 gpurun_out/s15/sparse.err
python3 -c "
import json; d=json.load(open('gpurun_out/s15/sparse.json')); r=d['roofline']
print('sparse fused: value %.4g ms/step %.3f'%(d['value'],d['ms_per_step']), {k:round(v['avg_launch_ms'],3) for k,v in r['per_picture_type'].items()})"
