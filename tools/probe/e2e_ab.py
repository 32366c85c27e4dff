#!/usr/bin/env python3
"""What a per-window callback costs the end-to-end figure: the timed gpu_parser run of bench.py's end_to_end with and
without bench.run_pipeline_sums (hold_last: the last windows kept and checksummed after the run), alternating, in one process.
  python tools/probe/e2e_ab.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import bench
import leon_ctypes as L
import stream_1080p

data = stream_1080p.load_varied()
kw = dict(parser_threads=16, gops_per_window=128, loop=960, gpu_parser=True, windows_in_flight=3)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    pipe = L.Pipeline(data, **kw)
    pipe.wait()
    st = pipe.stats()
    pipe.close()
    print("no callback      %.0f pictures/s" % (st["pictures"] / st["seconds"]), flush=True)
    pipe = L.Pipeline(data, on_window=lambda w, f: None, **kw)
    pipe.wait()
    st = pipe.stats()
    pipe.close()
    print("empty callback   %.0f pictures/s" % (st["pictures"] / st["seconds"]), flush=True)
    st, got, n = bench.run_pipeline_sums(L, torch, data, 0, 16, hold_last=3, **kw)
    print("last 3 windows   %.0f pictures/s (%d frames checked after the run)" % (st["pictures"] / st["seconds"], n), flush=True)
