#!/usr/bin/env python3
"""Which REGION's placement makes a launch position slow or fast?  (the same launch is up to 20 % slower or
faster from process to process on one box, stable for the life of its buffers, with every buffer physically contiguous.)
One process, bench.py's workload: time the eight launches of a step; then re-draw ONE region at a time -- the RGBA frames, the
coefficient arena, the slot ring (a new decoder) -- into a fresh contiguous allocation taken while the old one is still held, time
again, and so on in rounds.  The region whose re-draw moves a position is the one whose placement it is.
    python tools/probe/placement_probe.py [rounds] [--gops 128]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench
import numpy as np

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
gops = int(sys.argv[sys.argv.index("--gops") + 1]) if "--gops" in sys.argv else 128
contents, gop = bench.make_contents(8, bench.SEED)
import torch
import leon_ctypes as L
import synth as S
import shards

torch.cuda.set_device(0)
index = shards.make_index(bench.CW, bench.CH, bench.FW, bench.FH, rate_idx=3, n_gops=gops, gop_len=bench.GOP_LEN)
my = shards.shard_gops(index, 0, 1)
stream = torch.cuda.Stream()


def new_decoder():
    return L.Decoder(bench.CW, bench.CH, bench.FW, bench.FH, n_slots=gops * bench.GOP_LEN, device_id=0, stream=stream.cuda_stream, contiguous_slots=True)


dec = new_decoder()
wl = bench.Workload(L, S, dec, torch, contents, gop, my, bench.SEED, False, fused=True)
names = "I1 B1 P1 B2 P2 B3 P3 B4".split()


def measure(tag):
    d = wl.dec
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    d.timing_enable(True)
    d.timing_reset()
    for _ in range(6):
        wl.step()
    torch.cuda.synchronize()
    ls = [l["ms"] for l in d.timing_launches() if l["kind"] == 0]
    d.timing_reset()
    d.timing_enable(False)
    per = [sorted(ls[k::8])[len(ls[k::8]) // 2] for k in range(8)]
    where = {"rgba": [hex(b.ptr) for b in wl.arenas_rgba], "coef": hex(wl.arenas[0].ptr), "slots": hex(wl.dec.slot_device_ptr(0)[0])}
    print(json.dumps({"after": tag, "step_ms": round(sum(per), 3), **{n: round(v, 4) for n, v in zip(names, per)}, "where": where}), flush=True)


def rebuild():
    for li in range(len(wl.levels)):
        old = wl.batches[li]
        wl.batches[li] = wl.build_level(li)
        wl.dec.batch_destroy(old)


def redraw_rgba():
    old = list(wl.arenas_rgba)
    for li, b in enumerate(old):
        nb = L.DeviceBuffer(b.nbytes, 0)
        shape = tuple(wl.rgba_lv[li].shape)
        wl.rgba_lv[li] = nb.as_tensor(torch.uint8, shape)
        wl.arenas_rgba[li] = nb
    rebuild()
    for b in old:
        b.free()


def redraw_coef():
    old = wl.arenas[0]
    nb = L.DeviceBuffer(old.nbytes, 0)
    nb.as_tensor(torch.uint8, (old.nbytes,)).copy_(old.as_tensor(torch.uint8, (old.nbytes,)))
    torch.cuda.synchronize()
    for per_level in wl.level_tensors:
        for rec in per_level:
            d = rec[-1]
            for k, v in list(d.items()):
                d[k] = nb.as_tensor(v.dtype, tuple(v.shape), v.data_ptr() - old.ptr)
    wl.arenas[0] = nb
    rebuild()
    old.free()


def redraw_slots():
    global dec
    old = wl.dec
    for b in wl.batches:
        old.batch_destroy(b)
    nd = new_decoder()              # taken while the old ring is still held: another range of the pool / the driver
    wl.dec = nd
    wl.batches = [wl.build_level(li) for li in range(len(wl.levels))]
    old.close()
    dec = nd


measure("start")
if "--many-rings" in sys.argv:
    # ring after ring, every one HELD (so every one is a new range): step time against the ring's address
    held = []
    os.environ.pop("LEON_SLOT_ALIGN", None)
    os.environ.pop("LEON_SLOT_SKEW", None)
    for k in range(7):
        old = wl.dec
        for b in wl.batches:
            old.batch_destroy(b)
        held.append(old)
        wl.dec = new_decoder()
        wl.batches = [wl.build_level(li) for li in range(len(wl.levels))]
        measure("ring %d" % (k + 1))
    # and back to the first one
    for b in wl.batches:
        wl.dec.batch_destroy(b)
    held.append(wl.dec)
    wl.dec = held[0]
    wl.batches = [wl.build_level(li) for li in range(len(wl.levels))]
    measure("ring 0 again")
    rounds = 0
if "--slot-skews" in sys.argv:
    # the slot ring at chosen positions relative to a 64 MiB boundary, everything else held where it is
    for skew in [0, 2, 8, 10, 16, 32, 0, 10, 34, 1, 0]:
        os.environ["LEON_SLOT_ALIGN"] = str(64 << 20)
        os.environ["LEON_SLOT_SKEW"] = str(skew << 20)
        redraw_slots()
        measure("slots at 64 MiB + %d MiB" % skew)
    rounds = 0
for r in range(rounds):
    for tag, fn in (("rgba", redraw_rgba), ("coef", redraw_coef), ("slots", redraw_slots)):
        fn()
        measure("%s #%d" % (tag, r))
print(json.dumps({"pool": L.pool_stats()}))
