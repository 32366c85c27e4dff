"""CPU: the oracle against vectors produced by EXECUTING the reference's own pixel path --
its unmodified parser and GL driver (decoders/jsv.js), the shader text its own
composeShaders() assembles (decoders/jsv.js:2459-2470 over decoders/shaders/mpeg1video.js:18-29)
-- on the software WebGL machine of tools/softgl (tools/make_golden_glsl.js; build container
only).  This is the pin of dequantisation + IDCT (K1a/K1b) and of the forward-MC pass (K2) that
WebGL's absence used to leave open: no hand-typed shader arithmetic is on the fixture path.

    glsl_idct_cases.json  tensors -> jsv.prototype.IDCT_GL -> pass-1 scratch textures + planes
    glsl_streams.json     fixture streams -> the reference's decodeFrame loop -> planes (+ canvas)
"""
import base64
import hashlib
import os
import zlib

import numpy as np
import pytest

from helpers import ROOT, load_golden
from oracle import oracle_py as O

unz = lambda s, dt=np.uint8: np.frombuffer(zlib.decompress(base64.b64decode(s)), dtype=dt)
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_tensors(p):
    t = {"type": p["type"]}
    for k in ("coef_y", "coef_cb", "coef_cr", "mv_fwd"):
        if k in p:
            t[k] = unz(p[k], "<i2")
    for k in ("qscale", "intra", "repadd"):
        if k in p:
            t[k] = unz(p[k])
    return t


CASES = load_golden("glsl_idct_cases.json")["cases"]


def test_fixture_covers_the_adversarial_inputs():
    """the cases VERDICT r1 asked for are really in the data (checked on the inputs, not assumed)"""
    names = {c["name"] for c in CASES}
    assert {"synthetic_ippp_default_matrices", "full_int16_garbage_vectors_leave_picture",
            "single_coefficients_and_dc", "custom_intra_matrix_clamps_zero_to_one"} <= names
    g = next(c for c in CASES if c["name"].startswith("full_int16"))
    t = case_tensors(g["pictures"][1])
    assert t["coef_y"].min() < -30000 and t["coef_y"].max() > 30000          # |w| >= 65536 hand-off saturation
    cw, ch = g["coded_w"], g["coded_h"]
    mv = t["mv_fwd"].reshape(-1, 2)
    assert (np.abs(mv) > 62).any()                                             # vectors that leave the picture
    assert set(np.unique(t["repadd"])) >= {0, 127, 128, 255}                   # the > 0.5 threshold of RepAdd
    c = next(c for c in CASES if c["name"].startswith("custom_intra"))
    qm = unz(c["quant_matrices"])
    assert qm[:64].min() == 1 and qm[:64].max() == 255 and (qm[64:] == 16).all()
    t = case_tensors(c["pictures"][0])
    cw, ch = c["coded_w"], c["coded_h"]
    # an intra level of +1 with quantiser_scale 1 and a matrix entry < 8 floors to 0 -> the shader's 0 -> +1 step
    q = np.kron(t["qscale"].reshape(ch // 16, cw // 16), np.ones((16, 16), dtype=int))
    m = np.tile(qm[:64].reshape(8, 8), (ch // 8, cw // 8))
    lv = t["coef_y"].reshape(ch, cw).astype(int)
    ac = np.ones((ch, cw), dtype=bool)
    ac[::8, ::8] = False                                                       # intra DC takes another path
    assert (ac & (lv > 0) & (2 * lv * q * m < 16)).any()
    assert (ac & (2 * lv * q * m // 16 > 2047)).any() and (ac & (2 * lv * q * m // 16 < -2048)).any()   # both clamps


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_oracle_equals_reference_shaders(case):
    cw, ch = case["coded_w"], case["coded_h"]
    mbw = cw // 16
    qm = unz(case["quant_matrices"])
    pm = O.premultiplier()
    prev = None
    for i, p in enumerate(case["pictures"]):
        t = case_tensors(p)
        # pass 1: the idct_1d textures as the reference's FBOs hold them (RGBA8 (W/2) x H = int16 [H][W]);
        # the oracle's scratch is the same minus the plane-level vertical flip (window row 0 is the bottom)
        for comp, k in enumerate(("coef_y", "coef_cb", "coef_cr")):
            W, H = (cw, ch) if comp == 0 else (cw // 2, ch // 2)
            mine = O.pass1_plane(t[k], W, H, comp != 0, t["qscale"], t["intra"], mbw, qm, pm)
            theirs = unz(p["scratch"][comp], "<i2").reshape(H, W)[::-1]
            assert np.array_equal(mine, theirs), "%s picture %d pass-1 scratch of component %d" % (case["name"], i, comp)
        out = O.decode_picture(p["type"], cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                               repadd=t.get("repadd"), mv_fwd=t.get("mv_fwd"), qm=qm, ref_fwd=prev)
        ref = np.concatenate([unz(x) for x in p["planes"]])
        bad = np.nonzero(out != ref)[0]
        assert bad.size == 0, "%s picture %d (type %d): %d samples differ, first at %d" % (case["name"], i, p["type"], bad.size, bad[0])
        prev = ref.copy()     # the next P picture predicts from the REFERENCE's output, not from ours


STREAMS = load_golden("glsl_streams.json")["streams"]


@pytest.mark.parametrize("s", STREAMS, ids=lambda s: s["stream"])
def test_stream_end_to_end_equals_reference(s):
    """stream bytes -> the product's native front end (libleon_vlc.so) -> oracle  ==
       stream bytes -> the reference's parser -> the reference's IDCT_GL + shaders (softgl).
    For 352x240 (BASELINE config 1's stand-in) the fixture carries digests only."""
    import leon_vlc_ctypes as V
    data = open(os.path.join(ROOT, "tests", "golden", "streams", s["stream"]), "rb").read()
    st = V.Stream(data)
    cw, ch = s["coded_w"], s["coded_h"]
    qm = unz(s["custom_intra_matrix"])
    prev = None
    n = 0
    while True:
        p = st.next_picture(dense=True)
        if p is None:
            break
        r = s["pictures"][n]
        assert p["type"] == r["type"]
        for k, v in r["tensors_sha256"].items():
            assert sha(p[k]) == v, "picture %d: boundary tensor %s differs from what the reference uploaded" % (n, k)
        out = O.decode_picture(p["type"], cw, ch, p["coef_y"], p["coef_cb"], p["coef_cr"], p["qscale"], p["intra"],
                               repadd=p.get("repadd"), mv_fwd=p.get("mv_fwd"), qm=qm, ref_fwd=prev)
        assert [sha(x) for x in O.split_planes(out, cw, ch)] == r["planes_sha256"], "picture %d of %s" % (n, s["stream"])
        prev = out
        n += 1
    assert n == len(s["pictures"]) > 0


def test_canvas_of_renderframegl_close_to_oracle_gl_flavour():
    """renderFrameGL (player/easybits.player.js:2787-2858) + SHADER_FRAGMENT_YCBCRTORGBA
    (player/parts/end.js:77-156) executed on softgl vs the oracle's fp32 GL flavour.  Decision D10:
    the GL flavour is reported, not gated bit for bit (the order of the four products of `vec4 * mat4`
    is the GLSL compiler's choice); it must stay within 1 LSB, and equal on the crop geometry."""
    worst = 0
    for s in STREAMS:
        cw, ch, fw, fh = s["coded_w"], s["coded_h"], s["frame_w"], s["frame_h"]
        for r in s["pictures"]:
            if "canvas" not in r:
                continue
            y, cb, cr = (unz(x) for x in r["planes"])
            mine = O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "gl").astype(int)
            canvas = unz(r["canvas"]).reshape(fh, fw, 4)[::-1].astype(int)     # GL rows are bottom-up
            assert (canvas[..., 3] == 255).all()
            worst = max(worst, int(np.abs(mine - canvas).max()))
    assert worst <= 1, worst
