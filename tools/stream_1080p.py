#!/usr/bin/env python3
"""The synthetic 1080p IBBP stream that the end-to-end figures of bench.py and the 1080p tests of the GPU parser run on:
2 closed GOPs (24 pictures, 0.82 Mbit each), written by tools/jsv_writer.py from the seeded recipe of synth.py (seed
0x4C454F4E, the recipe of SURVEY.md 8d config 4).  Deterministic: the bytes are pinned by their SHA-256 below.  The file
(tools/probe/stream_1080p_2gop.bin, 2.4 MB) is git-ignored; `ensure()` writes it when it is absent -- 30 s of pure
Python -- so a clean clone has it after __graft_entry__.build(), and the GPU box gets it with the tree.

  python tools/stream_1080p.py            # write it if it is not there, print its path"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tools", "probe", "stream_1080p_2gop.bin")
SHA256 = "61f73c735dad3fd01b6d2579ef3b220548ee71d8e54b8c2d22b53fa3bba1f146"
BYTES = 2454299


def ensure(path=PATH):
    """path of the stream; written first if absent or damaged"""
    if os.path.exists(path) and os.path.getsize(path) == BYTES:
        return path
    for p in (os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import parse_bench
    tmp = path + ".tmp%d" % os.getpid()
    data = parse_bench.make_stream(2, tmp)
    if hashlib.sha256(data).hexdigest() != SHA256:
        os.unlink(tmp)
        raise RuntimeError("tools/stream_1080p.py: the writer no longer produces the pinned stream (synth.py or jsv_writer.py changed?)")
    os.replace(tmp, path)
    return path


def load():
    return open(ensure(), "rb").read()




# ---- the varied stream: N different GOPs, written by parallel processes and merged (jsv_writer.merge_gops) -------------
VARIED_GOPS = 16
VARIED_PATH = os.path.join(ROOT, "tools", "probe", "stream_1080p_%dgop_varied.bin" % VARIED_GOPS)


def _one_gop(g):
    for p in (os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import jsv_writer as W
    import synth as S
    rng = np.random.default_rng([0x4C454F4E, 1080, g])
    pics = []
    for ptype, disp, f, b in S.gop_ibbp(12):
        t = S.make_picture(rng, 1920, 1088, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
        t["display"] = disp
        pics.append(t)
    return W.write_stream(pics, 1920, 1088, 1920, 1080, gop_starts=[0])[0]


def ensure_varied(path=VARIED_PATH, n_gops=VARIED_GOPS, workers=None):
    """N closed IBBP GOPs with N different contents (GOP g is seeded by (0x4C454F4E, 1080, g)): what the end-to-end figures
    should be quoted on -- on the 2-GOP stream looped, a launch of the GPU parser holds every slice dozens of times, and
    lanes with identical slices do not diverge.  Written by `workers` processes (one GOP each, 15 s per GOP)."""
    if os.path.exists(path) and os.path.getsize(path) > n_gops * 1000000:
        return path
    import multiprocessing as mp
    for p in (os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import jsv_writer as W
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(workers or ncpu, n_gops))
    with mp.get_context("spawn").Pool(workers) as pool:
        gops = pool.map(_one_gop, range(n_gops))
    data, _ = W.merge_gops(gops, 1920, 1080)
    tmp = path + ".tmp%d" % os.getpid()
    open(tmp, "wb").write(data)
    os.replace(tmp, path)
    return path


def load_varied():
    return open(ensure_varied(), "rb").read()


if __name__ == "__main__":
    print(ensure())
    if "--varied" in sys.argv:
        print(ensure_varied())
