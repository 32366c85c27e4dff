"""Probe: wall time of the first 40 bench steps after a cold start (the device needs ~10 steps = 35 ms of this
workload to reach its steady clock: 4.07, 3.65, 3.40, 3.26, 3.20, 3.10, ... 3.03 ms)."""
import sys, os, time, json
sys.argv=['bench.py','--steps','1','--warmup','0','--no-cpu-baseline']
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import importlib
b=importlib.import_module('bench')
import leon_ctypes as L, synth as S, shards
index = shards.make_index(b.CW, b.CH, b.FW, b.FH, rate_idx=3, n_gops=48, gop_len=12)
stream=torch.cuda.Stream()
dec=L.Decoder(index["coded_w"], index["coded_h"], index["frame_w"], index["frame_h"], n_slots=48*12, device_id=0, stream=stream.cuda_stream)
batches, keep, host, gop = b.build_workload(L,S,dec,torch,48,seed=1)
levels=S.dependency_levels(gop)
level_slots=[np.array([g*12+e[1] for g in range(48) for e in lv],dtype=np.int32) for lv in levels]
rg=[torch.empty((len(s),b.FH,b.FW,4),dtype=torch.uint8,device='cuda') for s in level_slots]
def step():
    for k,bt in enumerate(batches):
        dec.batch_run(bt); dec.convert_rgba_batch(level_slots[k], rg[k].data_ptr())
ts=[]
for i in range(40):
    torch.cuda.synchronize(); t0=time.perf_counter(); step(); dec.sync(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print([round(t,2) for t in ts])
