#!/usr/bin/env python3
"""End to end on one GPU, PCIe included: stream bytes -> native front end (libleon_vlc, one stream
per GOP shard on its own host thread) -> sparse group lists staged from host memory
(leon_submit_sparse, LEON_MEM_HOST) -> reconstruction -> RGBA in device memory.

This is NOT the bench.py metric (that one starts with the boundary tensors resident in HBM); it is
the figure DESIGN.md quotes for the whole drop-in path.  The stream is one synthetic 1080p IBBP GOP
(tools/parse_bench.py writes and caches it) that every shard parses again and again.

  python tools/e2e_bench.py [--shards 16] [--seconds 10]"""
import argparse
import json
import os
import queue
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")]
import leon_ctypes as L          # noqa: E402  (imports torch first)
import leon_vlc_ctypes as V      # noqa: E402
import parse_bench               # noqa: E402

GOP = 12


def producer(data, q, stop):
    while not stop.is_set():
        st = V.Stream(data, threads=1)
        while not stop.is_set():
            p = st.next_picture()
            if p is None:
                break
            q.put(p)
        st.close()
    q.put(None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    a = ap.parse_args()
    import torch
    data = parse_bench.make_stream(1, "/tmp/leon_parse_bench_1.jsv")
    info = V.Stream(data, threads=1).info
    cw, ch, fw, fh = info.coded_width, info.coded_height, info.frame_width, info.frame_height
    n_slots = a.shards * 13
    dec = L.Decoder(cw, ch, fw, fh, n_slots=n_slots)
    rgba = torch.empty((fh, fw, 4), dtype=torch.uint8, device="cuda")
    stop = threading.Event()
    qs = [queue.Queue(maxsize=24) for _ in range(a.shards)]
    th = [threading.Thread(target=producer, args=(data, qs[i], stop), daemon=True) for i in range(a.shards)]
    for t in th:
        t.start()
    # per shard: ring position and the two anchors (prev_pic_framebuffer bookkeeping, jsv.js:665)
    state = [{"n": 0, "old": None, "new": None} for _ in range(a.shards)]
    done = 0
    entries = 0
    t0 = time.perf_counter()
    keep = []
    while time.perf_counter() - t0 < a.seconds:
        for i in range(a.shards):
            try:
                p = qs[i].get(timeout=1.0)
            except queue.Empty:
                continue
            if p is None:
                continue
            s = state[i]
            slot = i * 13 + s["n"] % 13
            s["n"] += 1
            if p["type"] == 1:
                s["old"], fwd, bwd = None, -1, -1
            elif p["type"] == 2:
                fwd, bwd = s["new"], -1
            else:
                fwd, bwd = (s["old"] if s["old"] is not None else s["new"]), s["new"]
            keep.clear()
            pic = L.make_sparse_picture(p["type"], slot, p["grp_off"], p["entries"], len(p["entries"]), p["qscale"], p["intra"],
                                        repadd=p["repadd"], mv_fwd=p["mv_fwd"], mv_bwd=p["mv_bwd"], mb_dir=p["mb_dir"],
                                        ref_fwd_slot=fwd, ref_bwd_slot=bwd, keep=keep)
            dec.submit_sparse([pic], L.MEM_HOST)
            dec.convert_rgba_batch(np.array([slot], np.int32), rgba.data_ptr())
            if p["type"] != 3:
                s["old"], s["new"] = s["new"], slot
            done += 1
            entries += len(p["entries"])
    dec.sync()
    dt = time.perf_counter() - t0
    stop.set()
    for q_ in qs:
        while not q_.empty():
            q_.get_nowait()
    print(json.dumps({"metric": "end-to-end 1080p pictures/s (parse + PCIe + reconstruct + RGBA), one GPU", "value": done / dt,
                      "macroblocks_per_s": done * 8160 / dt, "shards": a.shards, "seconds": dt, "pictures": done,
                      "entries_per_picture": entries / max(1, done), "host_threads": os.cpu_count()}))
    dec.close()


if __name__ == "__main__":
    main()
