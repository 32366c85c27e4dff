#!/usr/bin/env python3
"""Hunting a flake of round 3: in full-suite runs a [gpu-parser] pipeline test that follows its [host-parser] twin produced a
B picture with whole macroblocks wrong (2 of 3 runs), never alone.  This replays the sequences in one process: every frame
against the oracle; where frames differ, which macroblocks."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), ROOT]


def main():
    import torch
    import leon_ctypes as L
    import leon_vlc_ctypes as V
    from test_pipeline_gpu import ibbp_stream, oracle_frames, differing_macroblocks
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    streams = [("96x64 partial", ibbp_stream(96, 64, [6, 9, 3, 12, 6], seed=77), True),
               ("90x60 widths", ibbp_stream(96, 64, [12, 6], seed=21, frame=(90, 60)), False),
               ("61x45 widths", ibbp_stream(64, 48, [12, 6], seed=21, frame=(61, 45)), False)]
    wants = [oracle_frames(d) for _, d, _ in streams]
    bad = 0
    for rnd in range(rounds):
        if os.environ.get("LEON_STRESS_GARBAGE", "1") == "1":
            # memory with a history: a large physically contiguous buffer full of noise, freed -- what the pipelines below
            # allocate may be carved from it (ordinary allocations seem to come back zeroed, these do not)
            g = L.DeviceBuffer(int(os.environ.get("LEON_STRESS_GB", "2")) << 30)
            t = g.as_tensor(torch.uint8, (g.nbytes,))
            t.random_(1, 255)
            torch.cuda.synchronize()
            del t
            g.free()
        for (name, data, partial), want in zip(streams, wants):
            for gpu_parser in (False, True):
                got, lock = {}, threading.Lock()

                def on_window(window, frames):
                    with lock:
                        for f in frames:
                            got[(f["gop"], f["display_index"])] = L.read_frame(f)
                if partial:
                    offs = V.Stream(data, threads=1).keymap()
                    first = offs[1] + 3
                    buf = bytearray(len(data))
                    buf[:first] = data[:first]
                    pipe = L.Pipeline(bytes(buf), parser_threads=2, gops_per_window=1, gpu_parser=gpu_parser, on_window=on_window, valid_bytes=first)
                    time.sleep(0.05)
                    at = first
                    for step in (500, 1, 1800, 700, 10 ** 9):
                        n = min(step, len(data) - at)
                        pipe.feed(at + n, data[at:at + n], at)
                        at += n
                        if at == len(data):
                            break
                else:
                    pipe = L.Pipeline(data, parser_threads=2, gops_per_window=2, gpu_parser=gpu_parser, on_window=on_window)
                pipe.wait()
                pipe.close()
                for k in sorted(want):
                    if k not in got:
                        print("round %d %s gpu_parser=%s: frame %s missing" % (rnd, name, gpu_parser, k)); bad += 1
                    elif not np.array_equal(got[k].reshape(-1), np.asarray(want[k]).reshape(-1)):
                        print("round %d %s gpu_parser=%s: frame %s differs: %s" % (rnd, name, gpu_parser, k, differing_macroblocks(got[k], np.asarray(want[k]).reshape(got[k].shape))))
                        bad += 1
    print("rounds %d: %d bad frames" % (rounds, bad))


if __name__ == "__main__":
    main()
