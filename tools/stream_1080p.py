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


if __name__ == "__main__":
    print(ensure())
