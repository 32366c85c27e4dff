"""GPU: leon_pipeline_config.display_flavour = LEON_RGB_GL -- the pipeline's frames in the arithmetic of the display path the
reference actually runs: renderFrameGL drawing the three planes with SHADER_FRAGMENT_YCBCRTORGBA, fp32
(player/easybits.player.js:2787-2858, player/parts/end.js:77-156).  Held against
  * the `canvas` of tests/golden/glsl_streams.json = that shader EXECUTED (tools/softgl) on the planes the executed reference
    decoded from the same stream bytes: within 1 LSB (decision D10: the order of the four products of `vec4 * mat4` is the
    GLSL compiler's), alpha 255, crop geometry included;
  * the oracle's GL flavour on the oracle's planes: bit for bit, B pictures too (the reference drops them, jsv.js:613-616)."""
import os

import numpy as np
import pytest

from helpers import ROOT
from test_glsl_golden import STREAMS, unz
from test_pipeline_gpu import ibbp_stream, oracle_frames, run_pipeline

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import leon_ctypes
    leon_ctypes.load()
    return leon_ctypes


def oracle_gl_frames(data):
    """{(gop, display_index): RGBA in the GL flavour} from the oracle's planes; and the keys in decode order"""
    from oracle import oracle_py as O
    import leon_vlc_ctypes as V
    info = V.Stream(data, threads=1).info
    cw, ch, fw, fh = info.coded_width, info.coded_height, info.frame_width, info.frame_height
    detail = {}
    oracle_frames(data, detail)
    out = {}
    for k, d in detail.items():
        y, cb, cr = O.split_planes(d["planes"][:cw * ch * 3 // 2], cw, ch)
        out[k] = O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "gl")
    return out, list(detail)            # dicts keep insertion order = decode order


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_pipeline_frames_in_the_gl_flavour_against_the_executed_references_canvas(L, gpu_parser):
    worst, n = 0, 0
    for s in STREAMS:
        if not any("canvas" in r for r in s["pictures"]):
            continue
        data = open(os.path.join(ROOT, "tests", "golden", "streams", s["stream"]), "rb").read()
        want, keys = oracle_gl_frames(data)
        assert len(keys) == len(s["pictures"])                       # I / P streams: every picture of the fixture is a frame
        got, order, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2, gpu_parser=gpu_parser, display_flavour=L.RGB_GL)
        assert set(got) == set(want)
        fw, fh = s["frame_w"], s["frame_h"]
        for k, r in zip(keys, s["pictures"]):
            assert np.array_equal(got[k], want[k]), "%s frame %s: not the oracle's GL flavour" % (s["stream"], k)
            if "canvas" not in r:
                continue
            canvas = unz(r["canvas"]).reshape(fh, fw, 4)[::-1]       # GL rows are bottom-up
            assert got[k].shape == canvas.shape and (got[k][..., 3] == 255).all()
            worst = max(worst, int(np.abs(got[k].astype(int) - canvas.astype(int)).max()))
            n += 1
    assert n > 0 and worst <= 1, (n, worst)


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_gl_flavour_of_an_ibbp_stream_with_a_crop(L, gpu_parser):
    """B pictures take slots of their own on this road (they write planes: the conversion is a launch of its own); the frame
    is smaller than the coded picture and no multiple of 8 wide"""
    data = ibbp_stream(96, 64, [12, 5, 9], seed=99, frame=(90, 60))
    want, _ = oracle_gl_frames(data)
    got, order, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2, gpu_parser=gpu_parser, display_flavour=L.RGB_GL)
    assert set(got) == set(want) and stats["pictures"] == len(want)
    for k in want:
        assert got[k].shape == (60, 90, 4) and np.array_equal(got[k], want[k]), k
    twin = oracle_frames(data)
    assert any(not np.array_equal(twin[k], want[k]) for k in want)      # the two flavours are different arithmetic


def test_gl_flavour_refusals(L):
    data = open(os.path.join(ROOT, "tests", "golden", "streams", "yuva_ibbp_96x64.jsv"), "rb").read()
    with pytest.raises(L.LeonError, match="yuva"):
        L.Pipeline(data, display_flavour=L.RGB_GL)
    data = open(os.path.join(ROOT, "tests", "golden", "streams", "tiny_ip_32x32.jsv"), "rb").read()
    with pytest.raises(L.LeonError, match="display_flavour"):
        L.Pipeline(data, display_flavour=7)
    pipe = L.Pipeline(data, display_flavour=L.RGB_GL, gpu_parser=False)
    try:
        pipe.wait()
        assert pipe.info.display_flavour == L.RGB_GL
    finally:
        pipe.close()
