#!/usr/bin/env python3
"""Generate the synthetic .jsv fixture streams under tests/golden/streams/ (deterministic)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")]
import synth as S           # noqa: E402
import jsv_writer as W      # noqa: E402


def build(name, cw, ch, fw, fh, gops, seed, slice_mbs=None, **kw):
    rng = np.random.default_rng(seed)
    pics, starts = [], []
    for gop in gops:
        starts.append(len(pics))
        for ptype, disp, f, b in gop:
            force = 2 if (ptype == S.PIC_B and f is None) else None
            t = S.make_picture(rng, cw, ch, ptype, force_dir=force, **kw)
            t["display"] = disp
            pics.append(t)
    data, offs = W.write_stream(pics, cw, ch, fw, fh, gop_starts=starts, slice_mbs=slice_mbs, qm_intra=kw.get("qm_intra"))
    out = os.path.join(ROOT, "tests", "golden", "streams")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name + ".jsv"), "wb") as f:
        f.write(data)
    print(name, len(data), "bytes,", len(pics), "pictures, key map", offs)
    return pics


if __name__ == "__main__":
    build("tiny_ip_32x32", 32, 32, 32, 32, [S.gop_ippp(3)], 1)
    # BASELINE config 1 stand-in for the missing videos/leon.jsv: 352x240, 2 GOPs x (I + 11 P), key map
    build("leon_synth_352x240", 352, 240, 352, 240, [S.gop_ippp(12), S.gop_ippp(12)], 0x4C454F4E)
    # B pictures (beyond the reference parser, which drops them): product parser only
    build("ibbp_96x64", 96, 64, 90, 60, [S.gop_ibbp(12), S.gop_ibbp(6)], 7)
    # slices of 5 macroblocks (6 per row): mid-row starts and row-spanning slices, I + P only so that the
    # reference parser reads every picture
    build("slices5_ip_96x64", 96, 64, 96, 64, [S.gop_ippp(6)], 55, slice_mbs=5)
    # custom intra quantiser matrix in the sequence header (decoders/jsv.js:540-547; a custom NON-intra
    # matrix never reaches the reference's GPU path, :556, so it is left at the default)
    from make_glsl_cases import CUSTOM_INTRA
    build("custom_intra_ip_48x32", 48, 32, 48, 32, [S.gop_ippp(4)], 99, qm_intra=CUSTOM_INTRA.reshape(8, 8))
    # yuva: container flag `a` + four alpha blocks per macroblock (the repo's syntax, tools/jsv_writer.py); product parsers only
    build("yuva_ibbp_96x64", 96, 64, 96, 64, [S.gop_ibbp(9), S.gop_ippp(4)], 404, alpha=True)
