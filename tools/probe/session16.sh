#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s16
for c in 0 1 0 1; do
LEON_CONCURRENT_CLASSES=$c timeout -k 10 300 python bench.py --no-second-recipe --no-cpu-baseline --unique 2 > gpurun_out/s16/b$c.json 2> gpurun_out/s16/b$c.err; echo "rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/s16/b$c.json')); r=d['roofline']
print('concurrent=$c: value %.4g ms/step %.3f'%(d['value'],d['ms_per_step']), {k:round(v['avg_launch_ms'],3) for k,v in r['per_picture_type'].items()})"
done
