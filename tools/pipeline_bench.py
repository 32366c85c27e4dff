#!/usr/bin/env python3
"""End to end on one GPU through the NATIVE pipeline (include/leon_pipeline.h): pageable stream bytes ->
K parser threads (one libleon_vlc stream per GOP shard, output written straight into pinned memory) ->
one upload per GOP -> one launch per picture type and dependency level across a window of GOPs, display
conversion fused in -> RGBA frames in device memory, delivered by callback.  No interpreter in the loop:
Python only starts the pipeline and waits.

This is NOT the bench.py metric (that one starts with the boundary tensors resident in HBM); it is the
figure DESIGN.md quotes for the whole drop-in path.  The stream is a synthetic 1080p IBBP stream of
--gops GOPs (tools/parse_bench.py writes and caches it; tools/probe/stream_1080p_<n>gop.bin is used when
present) decoded --loop times over.

  python tools/pipeline_bench.py [--gops 2] [--loop 64] [--threads 16] [--window 32]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gops", type=int, default=2)
    ap.add_argument("--loop", type=int, default=64)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--window", type=int, default=32)
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--gpu-parser", action="store_true", help="decode the slice layer on the GPU (leon_pipeline_config.gpu_parser)")
    ap.add_argument("--varied", action="store_true", help="the 16-GOP stream with 16 different contents (tools/stream_1080p.py) instead of --gops GOPs")
    a = ap.parse_args()
    cached = os.path.join(ROOT, "tools", "probe", "stream_1080p_%dgop.bin" % a.gops)
    if a.varied:
        import stream_1080p
        data = stream_1080p.load_varied()
        a.gops = stream_1080p.VARIED_GOPS
    elif os.path.exists(cached):
        data = open(cached, "rb").read()
    else:
        import parse_bench
        data = parse_bench.make_stream(a.gops, "/tmp/leon_parse_bench_%d.jsv" % a.gops)
    import ctypes
    import leon_ctypes as L
    L.load()

    def free_device_bytes():                      # what the pipeline holds on the device = free before - free after its creation
        hip = ctypes.CDLL("libamdhip64.so")
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        return f.value if hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0 else None
    free0 = free_device_bytes()
    t0 = time.perf_counter()
    pipe = L.Pipeline(data, parser_threads=a.threads, gops_per_window=a.window, windows_in_flight=a.inflight, loop=a.loop, gpu_parser=a.gpu_parser)
    free1 = free_device_bytes()
    pipe.wait()
    wall = time.perf_counter() - t0
    s = pipe.stats()
    pipe.close()
    mbs = (pipe.info.coded_width // 16) * (pipe.info.coded_height // 16)
    print(json.dumps({
        "metric": "end-to-end %dx%d pictures/s (parse + PCIe + reconstruct + RGBA in device memory), native pipeline, one GPU"
                  % (pipe.info.frame_width, pipe.info.frame_height),
        "value": s["pictures"] / s["seconds"], "macroblocks_per_s": s["pictures"] * mbs / s["seconds"],
        "pictures": s["pictures"], "seconds": s["seconds"], "wall_seconds_incl_setup": wall, "windows": s["windows"],
        "slice_layer": "GPU (csrc/leon_vlc_gpu.h)" if a.gpu_parser else "host threads (libleon_vlc.so)",
        "device_gb_held_by_the_pipeline": (free0 - free1) / 1e9 if free0 is not None and free1 is not None else None,
        "parser_threads": pipe.info.parser_threads, "gops_per_window": pipe.info.gops_per_window,
        "parse_seconds_summed_over_threads": s["parse_seconds_sum"],
        "parser_pictures_per_s_per_thread": s["pictures"] / s["parse_seconds_sum"] if s["parse_seconds_sum"] else None,
        "upload_gb": s["upload_bytes"] / 1e9, "upload_gbps": s["upload_bytes"] / 1e9 / s["seconds"],
        "entries_per_picture": s["entries"] / max(1, s["pictures"]), "stream_bytes": s["stream_bytes"],
        "stream_megabit_per_picture": s["stream_bytes"] * 8 / 1e6 / (12 * a.gops), "host_threads": os.cpu_count(),
        "stream": "%d different GOPs, looped %d times" % (a.gops, a.loop)}))


if __name__ == "__main__":
    main()
