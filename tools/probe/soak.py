"""Probe: create / use / destroy decoders of several sizes through both boundaries; prints the device-memory delta."""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mpeg1video-decoder-webgl_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import leon_ctypes as L, synth as S
from helpers import hip_submit, hip_submit_sparse, planes_flat
rng=np.random.default_rng(5)
free0=torch.cuda.mem_get_info()[0]
ref=None
import os
N=int(os.environ.get('SOAK_N','60'))
for it in range(N):
    cw,ch=[(352,240),(96,64),(1920,1088),(176,144)][it%4]
    dec=L.Decoder(cw,ch,n_slots=6)
    keep=[]
    gop=[(S.PIC_I,0,None,None),(S.PIC_P,2,0,None),(S.PIC_B,1,0,2)]
    for ptype,slot,f,b in gop:
        t=S.make_picture(rng,cw,ch,ptype) if it<8 or it%4!=2 else S.make_picture(np.random.default_rng(1),cw,ch,ptype)
        t['slot']=slot; t['ref_fwd']=f; t['ref_bwd']=b
        (hip_submit if it%2 else hip_submit_sparse)(L,dec,t,keep,*(() if it%2 else (cw,ch)))
    dec.convert_rgba(1)
    dec.sync(); dec.close()
free1=torch.cuda.mem_get_info()[0]
print('device memory delta MB', (free0-free1)/1e6)
