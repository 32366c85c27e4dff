#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR -d gpurun_out/s5/valu --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe --unique 2 > gpurun_out/s5/valu.log 2>&1 || echo "valu failed"
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS -d gpurun_out/s5/wait --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe --unique 2 > gpurun_out/s5/wait.log 2>&1 || echo "wait failed"
echo done
