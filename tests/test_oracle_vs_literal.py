"""CPU: two independent restatements of the GLSL path must agree bit for bit --
oracle/leon_oracle.c (image-space integers) vs oracle/glsl_literal.py (the emulated GL
machine: textures, normalised coordinates, fragments, float32 where the shader says float)."""
import numpy as np
import pytest

from oracle import glsl_literal as G
from oracle import oracle_py as O


def _rnd_coef(rng, W, H, dens, amp):
    c = rng.integers(-amp, amp + 1, size=(H, W)).astype(np.int16)
    c[rng.random((H, W)) > dens] = 0
    return c


@pytest.mark.parametrize("amp,dens,custom", [(60, 0.3, False), (60, 0.3, True), (700, 0.9, False), (32767, 0.5, True)])
def test_i_and_p_pictures(amp, dens, custom):
    rng = np.random.default_rng(amp + custom)
    cw, ch = 48, 32
    mbw, mbh = cw // 16, ch // 16
    qm = O.default_qm()
    pm = O.premultiplier()
    if custom:                       # small entries reach the floor()==0 -> +1 quirk
        qm = qm.copy()
        qm[:64] = rng.integers(1, 40, size=64)
    for trial in range(2):
        coef = [_rnd_coef(rng, cw, ch, dens, amp), _rnd_coef(rng, cw // 2, ch // 2, dens, amp),
                _rnd_coef(rng, cw // 2, ch // 2, dens, amp)]
        qs = rng.integers(1, 32, size=mbw * mbh).astype(np.uint8)
        ia = (rng.random(mbw * mbh) < 0.5).astype(np.uint8) * 255
        got = O.split_planes(O.decode_picture(1, cw, ch, *coef, qs, ia, qm=qm, pm=pm), cw, ch)
        exp = G.decode_picture_literal(1, cw, ch, coef, qs, ia, None, None, qm, pm, None)
        for a, b in zip(got, exp):
            assert np.array_equal(a, b)
        prev = [rng.integers(0, 256, size=p.shape).astype(np.uint8) for p in got]
        mv = rng.integers(-9, 10, size=mbh * mbw * 2).astype(np.int16)      # some leave the picture
        rep = (rng.random(mbw * mbh) < 0.2).astype(np.uint8) * 255
        ia2 = np.where(rep > 0, 255, ia).astype(np.uint8)
        ref = np.concatenate([p.ravel() for p in prev])
        got = O.split_planes(O.decode_picture(2, cw, ch, *coef, qs, ia2, repadd=rep, mv_fwd=mv, qm=qm, pm=pm,
                                              ref_fwd=ref), cw, ch)
        exp = G.decode_picture_literal(2, cw, ch, coef, qs, ia2, rep, mv, qm, pm, prev)
        for a, b in zip(got, exp):
            assert np.array_equal(a, b)


def test_handoff_store_saturation():
    """_B()/_E() through an RGBA8 render target for every interesting magnitude."""
    for w in [0, 1, -1, 32767, 32768, -32768, -32769, 65535, 65536, 65537, -65536, -65537, 100000, -100000, 1 << 20]:
        lo, hi = G._B(np.float32(w))
        b = G._unorm8(np.array([lo, hi], dtype=np.float32)).astype(np.float32) / np.float32(255.0)
        exp = int(G._E(b[0], b[1]))
        assert O.lib().lo_handoff_store(w) == exp, w
