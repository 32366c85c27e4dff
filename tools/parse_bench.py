#!/usr/bin/env python3
"""Throughput of the native bitstream front end (libleon_vlc.so) on a synthetic 1080p IBBP stream,
by worker-thread count, next to the product's JavaScript parser (the mirror of the reference's
bit-serial parser) on the same bytes.  The stream is written once by tools/jsv_writer.py
(slow, pure Python) and cached under /tmp.

  python tools/parse_bench.py [--gops 1] [--threads 1,2,4,8,16]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")]
import synth as S           # noqa: E402
import jsv_writer as W      # noqa: E402
import leon_vlc_ctypes as V  # noqa: E402

CW, CH, FW, FH = 1920, 1088, 1920, 1080


def make_stream(gops, path):
    if os.path.exists(path):
        return open(path, "rb").read()
    rng = np.random.default_rng(0x4C454F4E)
    pics, starts = [], []
    for _ in range(gops):
        starts.append(len(pics))
        for ptype, disp, f, b in S.gop_ibbp(12):
            force = 2 if (ptype == S.PIC_B and f is None) else None
            t = S.make_picture(rng, CW, CH, ptype, force_dir=force)
            t["display"] = disp
            pics.append(t)
    data, _ = W.write_stream(pics, CW, CH, FW, FH, gop_starts=starts)
    with open(path, "wb") as f:
        f.write(data)
    return data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gops", type=int, default=1)
    ap.add_argument("--threads", default="1,2,4,8,16")
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--no-js", action="store_true")
    a = ap.parse_args()
    t0 = time.perf_counter()
    data = make_stream(a.gops, "/tmp/leon_parse_bench_%d.jsv" % a.gops)
    out = {"stream_bytes": len(data), "pictures": 12 * a.gops, "write_s": round(time.perf_counter() - t0, 1), "native": {}}
    lib = V.load()
    import ctypes as C
    for nt in [int(x) for x in a.threads.split(",")]:
        best = None
        for _ in range(a.repeat):
            st = V.Stream(data, threads=nt)
            p = V.Picture()
            n = entries = 0
            t0 = time.perf_counter()
            while lib.leon_vlc_next_picture(st.h, C.byref(p)) == 1:       # raw call: no numpy copies in the timing
                n += 1
                entries += p.n_entries
            dt = time.perf_counter() - t0
            st.close()
            best = dt if best is None else min(best, dt)
        out["native"][str(nt)] = {"pictures_per_s": n / best, "mbit_per_s": len(data) * 8 / best / 1e6,
                                  "macroblocks_per_s": n * 8160 / best, "entries_per_picture": entries / n}
    if not a.no_js:
        js = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js", "jsv_decoder.js")
        code = """
          const {JsvDecoder} = require(%r); const fs = require('fs');
          const bytes = new Uint8Array(fs.readFileSync(%r));
          let best = 1e9, n = 0;
          for (let rep = 0; rep < 3; rep++) {
            const d = new JsvDecoder({}); n = 0; d.on('frame', () => n++);
            d.addBuffer(bytes); d._initMeta();
            const t0 = process.hrtime.bigint(); while (d.decodeFrame()); 
            best = Math.min(best, Number(process.hrtime.bigint() - t0) / 1e9);
          }
          console.log(JSON.stringify({n, s: best}));""" % (js, "/tmp/leon_parse_bench_%d.jsv" % a.gops)
        r = subprocess.run(["node", "-e", code], capture_output=True, text=True, timeout=1200)
        if r.returncode == 0:
            j = json.loads(r.stdout)
            out["javascript_1_thread"] = {"pictures_per_s": j["n"] / j["s"], "macroblocks_per_s": j["n"] * 8160 / j["s"]}
        else:
            out["javascript_1_thread"] = {"error": r.stderr[-300:]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
