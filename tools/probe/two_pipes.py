#!/usr/bin/env python3
"""One pipeline against TWO on the same GPU (the stream's GOPs split g = r mod 2, leon_pipeline_config.shard_index / shard_count),
in one process: does a single pipeline leave the device idle?   python tools/probe/two_pipes.py [window] [loop]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import leon_ctypes as L
import stream_1080p

data = stream_1080p.load_varied()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LOOP = int(sys.argv[2]) if len(sys.argv) > 2 else 960


def run(n, threads):
    pipes = [L.Pipeline(data, parser_threads=threads, gops_per_window=W, loop=LOOP * n // 1, gpu_parser=True, windows_in_flight=3,
                        shard_index=r, shard_count=n) if n > 1 else
             L.Pipeline(data, parser_threads=threads, gops_per_window=W, loop=LOOP, gpu_parser=True, windows_in_flight=3) for r in range(n)]
    t0 = time.perf_counter()
    for p in pipes:
        p.wait()
    wall = time.perf_counter() - t0
    st = [p.stats() for p in pipes]
    for p in pipes:
        p.close()
    pics = sum(s["pictures"] for s in st)
    return pics, max(s["seconds"] for s in st), [round(s["pictures"] / s["seconds"]) for s in st]


for rep in range(2):
    pics, secs, rates = run(1, 16)
    print("1 pipeline   %7.0f pictures/s  (%d pictures in %.2f s)" % (pics / secs, pics, secs), flush=True)
    for n in (2, 3, 4):
        pics, secs, rates = run(n, max(2, 16 // n))
        print("%d pipelines  %7.0f pictures/s  (%d pictures, slowest %.2f s; each %s)" % (n, pics / secs, pics, secs, rates), flush=True)
