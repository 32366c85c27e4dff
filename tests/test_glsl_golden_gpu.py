"""GPU: libleon_hip.so (through the C ABI) against the vectors produced by executing the
reference's own shader text on tools/softgl (tests/golden/glsl_*.json, see
tests/test_glsl_golden.py).  Bit-exact, dense and sparse boundary, and both launch forms: the plain
reconstruction kernel (k_recon) and the one with the display conversion fused in (k_recon_display, what
bench.py and the pipeline run): its planes against the same fixtures, its RGBA frame against the oracle's
CPU twin of the reference's YCbCrToRGBA applied to the fixture's planes."""
import os

import numpy as np
import pytest

from helpers import ROOT, hip_submit, hip_submit_sparse, planes_flat
from test_glsl_golden import CASES, STREAMS, case_tensors, sha, unz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import leon_ctypes
    leon_ctypes.load()
    return leon_ctypes


def split(flat, cw, ch):
    n = cw * ch
    return flat[:n].reshape(ch, cw), flat[n:n + n // 4].reshape(ch // 2, cw // 2), flat[n + n // 4:].reshape(ch // 2, cw // 2)


def rgba_twin(flat, cw, ch):
    from oracle import oracle_py as O
    return O.ycbcr_to_rgba(*O.split_planes(flat, cw, ch), cw, cw, ch, "cpu")


@pytest.mark.parametrize("fused", [False, True], ids=["k_recon", "k_recon_display"])
@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_hip_equals_reference_shaders(L, case, sparse, fused):
    import torch
    cw, ch = case["coded_w"], case["coded_h"]
    qm = unz(case["quant_matrices"])
    dec = L.Decoder(cw, ch, n_slots=4)
    frame = torch.zeros((ch, cw, 4), dtype=torch.uint8, device="cuda") if fused else None
    rgba_out = frame.data_ptr() if fused else None
    try:
        dec.set_quant_matrices(qm[:64], qm[64:])
        keep = []
        prev = None
        for i, p in enumerate(case["pictures"]):
            t = case_tensors(p)
            t["slot"] = 1
            t["ref_fwd"] = None
            if p["type"] == 2:
                # predict from the REFERENCE's previous output, so one differing picture cannot hide the next
                dec.write_planes(0, *split(prev, cw, ch))
                t["ref_fwd"] = 0
            (hip_submit_sparse(L, dec, t, keep, cw, ch, rgba_out) if sparse else hip_submit(L, dec, t, keep, rgba_out))
            got = planes_flat(*dec.read_planes(1))
            ref = np.concatenate([unz(x) for x in p["planes"]])
            bad = np.nonzero(got != ref)[0]
            assert bad.size == 0, "%s picture %d (type %d): %d samples differ, first at %d (got %d want %d)" % (
                case["name"], i, p["type"], bad.size, bad[0], got[bad[0]], ref[bad[0]])
            if fused:
                assert np.array_equal(frame.cpu().numpy(), rgba_twin(ref, cw, ch)), "%s picture %d: RGBA frame" % (case["name"], i)
            prev = ref
    finally:
        dec.close()


@pytest.mark.parametrize("fused", [False, True], ids=["k_recon", "k_recon_display"])
@pytest.mark.parametrize("s", STREAMS, ids=lambda s: s["stream"])
def test_stream_through_native_front_end_and_hip(L, s, fused):
    """stream bytes -> libleon_vlc.so -> leon_submit_sparse -> planes == the reference's own
    decode of the same bytes (its parser, its IDCT_GL, its shaders on softgl)."""
    import leon_vlc_ctypes as V
    data = open(os.path.join(ROOT, "tests", "golden", "streams", s["stream"]), "rb").read()
    st = V.Stream(data)
    cw, ch = s["coded_w"], s["coded_h"]
    qm = unz(s["custom_intra_matrix"])
    import torch
    dec = L.Decoder(cw, ch, n_slots=4)
    frame = torch.zeros((ch, cw, 4), dtype=torch.uint8, device="cuda") if fused else None
    try:
        dec.set_quant_matrices(qm[:64], qm[64:])
        keep = []
        n = 0
        while True:
            p = st.next_picture(dense=True)
            if p is None:
                break
            r = s["pictures"][n]
            t = dict(p)
            t["slot"] = n & 1
            t["ref_fwd"] = None if p["type"] == 1 else (n & 1) ^ 1
            hip_submit_sparse(L, dec, t, keep, cw, ch, frame.data_ptr() if fused else None)
            planes = dec.read_planes(n & 1)
            got = [sha(x) for x in planes]
            assert got == r["planes_sha256"], "picture %d of %s" % (n, s["stream"])
            if fused:
                assert np.array_equal(frame.cpu().numpy(), rgba_twin(planes_flat(*planes), cw, ch)), "picture %d of %s: RGBA frame" % (n, s["stream"])
            n += 1
        assert n == len(s["pictures"])
    finally:
        dec.close()


def test_gl_flavour_kernel_against_the_executed_references_canvas(L):
    """the display path the reference actually runs -- renderFrameGL + SHADER_FRAGMENT_YCBCRTORGBA, fp32
    (player/easybits.player.js:2787-2858, player/parts/end.js:77-156) -- executed on softgl: the `canvas` of
    tests/golden/glsl_streams.json.  leon_convert_rgba(LEON_RGB_GL) (k_rgba_gl) on the same fixture planes stays within
    1 LSB of it (decision D10: the order of the four products of `vec4 * mat4` is the GLSL compiler's), alpha 255, on the
    crop geometry (frame size < coded size included); and equals the oracle's GL flavour bit for bit."""
    from oracle import oracle_py as O
    worst, n = 0, 0
    for s in STREAMS:
        cw, ch, fw, fh = s["coded_w"], s["coded_h"], s["frame_w"], s["frame_h"]
        dec = L.Decoder(cw, ch, fw, fh, n_slots=2)
        try:
            for r in s["pictures"]:
                if "canvas" not in r:
                    continue
                y, cb, cr = (unz(x) for x in r["planes"])
                dec.write_planes(1, y.reshape(ch, cw), cb.reshape(ch // 2, cw // 2), cr.reshape(ch // 2, cw // 2))
                got = dec.convert_rgba(1, L.RGB_GL)
                canvas = unz(r["canvas"]).reshape(fh, fw, 4)[::-1]              # GL rows are bottom-up
                assert got.shape == canvas.shape and (got[..., 3] == 255).all()
                worst = max(worst, int(np.abs(got.astype(int) - canvas.astype(int)).max()))
                assert np.array_equal(got, O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "gl")), "k_rgba_gl differs from the oracle's GL flavour"
                n += 1
        finally:
            dec.close()
    assert n > 0 and worst <= 1, (n, worst)
