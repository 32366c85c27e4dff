"""CPU: the oracle against the fixtures generated from the reference's own JavaScript
(tools/make_golden.js, run under Node in the build container) -- SURVEY.md 8c."""
import numpy as np

from helpers import b64, load_golden
from oracle import oracle_py as O


def test_mc_predictor_matches_copymacroblock():
    """722 vectors x (Y, Cb, Cr) from jsv.prototype.copyMacroblock (decoders/jsv.js:895-1129)."""
    g = load_golden("mc_copymacroblock.json")
    cw, ch = g["coded_w"], g["coded_h"]
    mbw = cw // 16
    ry, rcb, rcr = b64(g["ref_y"]), b64(g["ref_cb"]), b64(g["ref_cr"])
    assert len(g["cases"]) == 722
    for c in g["cases"]:
        mv = np.zeros((ch // 16) * mbw * 2, dtype=np.int16)
        mb = c["mbRow"] * mbw + c["mbCol"]
        mv[2 * mb], mv[2 * mb + 1] = c["mvH"], c["mvV"]
        py = O.predict_plane(ry, cw, ch, 0, mv, mbw)
        pcb = O.predict_plane(rcb, cw // 2, ch // 2, 1, mv, mbw)
        pcr = O.predict_plane(rcr, cw // 2, ch // 2, 1, mv, mbw)
        r, q = c["mbRow"], c["mbCol"]
        assert np.array_equal(py[16 * r:16 * r + 16, 16 * q:16 * q + 16].ravel(), b64(c["y"])), (c["mvH"], c["mvV"])
        assert np.array_equal(pcb[8 * r:8 * r + 8, 8 * q:8 * q + 8].ravel(), b64(c["cb"])), (c["mvH"], c["mvV"])
        assert np.array_equal(pcr[8 * r:8 * r + 8, 8 * q:8 * q + 8].ravel(), b64(c["cr"])), (c["mvH"], c["mvV"])


def test_rgba_matches_ycbcrtorgba():
    """jsv.prototype.YCbCrToRGBA (player/easybits.player.js:2674-2785): random planes, an odd
    crop (the reference's index drift included) and every exact decimal rounding tie."""
    g = load_golden("rgb_ycbcrtorgba.json")
    assert g["n_ties"] > 100
    for s in g["sets"]:
        cw, fw, fh = s["coded_w"], s["frame_w"], s["frame_h"]
        out = O.ycbcr_to_rgba(b64(s["y"]), b64(s["cb"]), b64(s["cr"]), cw, fw, fh, "cpu")
        assert np.array_equal(out, b64(s["rgba"]).reshape(fh, fw, 4)), s["name"]


def test_gl_flavour_close_to_cpu_twin():
    """The GL matrix (player/parts/end.js:87-92) and the CPU twin fold constants differently:
    they agree within 2 LSB (reported, not gated on the GPU: decision D10)."""
    rng = np.random.default_rng(0)
    y = rng.integers(0, 256, (32, 32)).astype(np.uint8)
    cb = rng.integers(0, 256, (16, 16)).astype(np.uint8)
    cr = rng.integers(0, 256, (16, 16)).astype(np.uint8)
    a = O.ycbcr_to_rgba(y, cb, cr, 32, 32, 32, "cpu").astype(int)
    b = O.ycbcr_to_rgba(y, cb, cr, 32, 32, 32, "gl").astype(int)
    assert np.abs(a - b).max() <= 2
