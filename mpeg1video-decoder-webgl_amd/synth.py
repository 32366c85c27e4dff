"""Synthetic boundary tensors for tests and bench.py.

Produces exactly what the reference's slice loop leaves behind for IDCT_GL
(decoders/jsv.js:1177-1336): dense int16 coefficient planes of raw quantised
levels, the quantiser-scale / intra / RepAdd maps and the motion-vector map --
from a real forward DCT of bounded 8-bit content, so that the int16 hand-off of
pass 1 cannot overflow (decision D6).  numpy only; no oracle, no reference.
"""
import numpy as np

DEFAULT_INTRA_QUANT = np.array([
    8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83], dtype=np.uint8)
DEFAULT_NON_INTRA_QUANT = np.full(64, 16, dtype=np.uint8)

PIC_I, PIC_P, PIC_B = 1, 2, 3


def _dct_matrix():
    k = np.arange(8)[:, None]
    n = np.arange(8)[None, :]
    d = np.cos((2 * n + 1) * k * np.pi / 16.0) * 0.5
    d[0, :] *= 1.0 / np.sqrt(2.0)
    return d.astype(np.float32)


_D = _dct_matrix()


def fdct_plane(plane):
    """plane [H][W] float32 -> coefficients [H/8][W/8][8][8] (orthonormal DCT-II: DC = 8*mean)."""
    H, W = plane.shape
    b = plane.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3)
    return np.einsum('ki,rqij,lj->rqkl', _D, b, _D, optimize=True)


def blocks_to_plane(levels):
    """[H/8][W/8][8][8] -> [H][W] (each block at its pixel position, natural order)."""
    R, Q = levels.shape[:2]
    return levels.transpose(0, 2, 1, 3).reshape(R * 8, Q * 8)


def smooth_scene(rng, W, H, lo=16.0, hi=235.0, noise=8.0):
    """Sum of 4 random 2-D cosines + uniform noise."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    s = np.zeros((H, W), dtype=np.float32)
    for _ in range(4):
        fx, fy = rng.uniform(0.002, 0.05, size=2)
        ph = rng.uniform(0, 2 * np.pi)
        s += np.cos(2 * np.pi * (fx * x + fy * y) + ph).astype(np.float32)
    s = (s / 4.0 + 1.0) * 0.5 * (hi - lo) + lo
    s += rng.uniform(-noise, noise, size=(H, W)).astype(np.float32)
    return np.clip(s, 0, 255)


def _mb_to_blocks(arr_mb, mbw, mbh, chroma):
    a = np.asarray(arr_mb).reshape(mbh, mbw)
    return a if chroma else np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)


def quantise_plane(coefs, qscale_mb, intra_mb, mbw, mbh, chroma, qm_intra, qm_non):
    """MPEG-1 style forward quantiser matching the reference's dequantiser
    (decoders/shaders/mpeg1video.js:22): intra AC l = trunc(8c/(q*Q)), intra DC =
    round(c/8) in 0..255, non-intra l = trunc(8c/(q*Q)) (dead zone)."""
    q = _mb_to_blocks(qscale_mb, mbw, mbh, chroma).astype(np.float32)[:, :, None, None]
    ia = _mb_to_blocks(intra_mb, mbw, mbh, chroma).astype(bool)[:, :, None, None]
    Qi = qm_intra.reshape(8, 8).astype(np.float32)[None, None]
    Qn = qm_non.reshape(8, 8).astype(np.float32)[None, None]
    li = np.trunc(8.0 * coefs / (q * Qi))
    ln = np.trunc(8.0 * coefs / (q * Qn))
    lv = np.where(ia, li, ln)
    dc = np.clip(np.rint(coefs[..., 0, 0] / 8.0), 0, 255)
    lv[..., 0, 0] = np.where(ia[..., 0, 0], dc, lv[..., 0, 0])
    return np.clip(lv, -255, 255).astype(np.int16)


def clip_vectors(mv, mbw, mbh, cw, ch, margin=1):
    """Keep every predictor window inside the coded picture (half-pel units)."""
    mv = mv.reshape(mbh, mbw, 2).astype(np.int32)
    mx = np.arange(mbw)[None, :]
    my = np.arange(mbh)[:, None]
    mv[..., 0] = np.clip(mv[..., 0], -32 * mx, 2 * cw - 2 * margin - 32 * (mx + 1))
    mv[..., 1] = np.clip(mv[..., 1], -32 * my, 2 * ch - 2 * margin - 32 * (my + 1))
    return mv.reshape(-1).astype(np.int16)


def _vectors(rng, mbw, mbh, mv_range, coherent):
    """per-macroblock (H, V) vectors, uniform in +-mv_range half-pel; coherent = k > 1: one vector
    per k x k macroblocks (a motion field, the way real content moves) instead of one per macroblock"""
    if coherent <= 1:
        return rng.integers(-mv_range, mv_range + 1, size=mbw * mbh * 2).astype(np.int16)
    gh, gw = -(-mbh // coherent), -(-mbw // coherent)
    coarse = rng.integers(-mv_range, mv_range + 1, size=(gh, gw, 2)).astype(np.int16)
    return np.repeat(np.repeat(coarse, coherent, axis=0), coherent, axis=1)[:mbh, :mbw].reshape(-1).copy()


def make_picture(rng, cw, ch, ptype, mv_range=31, in_picture=True, qm_intra=DEFAULT_INTRA_QUANT,
                 qm_non=DEFAULT_NON_INTRA_QUANT, intra_frac=0.10, skip_frac=0.15, uncoded_frac=0.3,
                 force_dir=None, mv_coherent=1, alpha=False):
    """Boundary tensors of one picture.  Returns a dict of numpy arrays.
    alpha: a fourth, luma-sized component ("coef_a") for yuva streams (container flag `a`, decoders/jsv.js:256-259)."""
    mbw, mbh = cw // 16, ch // 16
    nmb = mbw * mbh
    qscale = ((np.arange(mbh)[:, None] % 30) + 2 + np.zeros((1, mbw), dtype=np.int64)).astype(np.uint8).reshape(-1)
    t = {"type": ptype, "qscale": qscale}
    if ptype == PIC_I:
        intra = np.full(nmb, 255, dtype=np.uint8)
        skipped = np.zeros(nmb, dtype=bool)
    else:
        u = rng.random(nmb)
        intra = np.where(u < intra_frac, 255, 0).astype(np.uint8)
        skipped = (u >= intra_frac) & (u < intra_frac + skip_frac)
        t["repadd"] = intra.copy()
        mv = _vectors(rng, mbw, mbh, mv_range, mv_coherent)
        mvb = _vectors(rng, mbw, mbh, mv_range, mv_coherent)
        if ptype == PIC_P:
            mv.reshape(-1, 2)[skipped] = 0           # skipped P macroblocks reset the vector (jsv.js:754-778)
        if in_picture:
            mv = clip_vectors(mv, mbw, mbh, cw, ch)
            mvb = clip_vectors(mvb, mbw, mbh, cw, ch)
        t["mv_fwd"] = mv
        if ptype == PIC_B:
            t["mv_bwd"] = mvb
            d = rng.integers(0, 4, size=nmb)          # fwd : bwd : bi = 1 : 1 : 2
            t["mb_dir"] = np.where(d == 0, 1, np.where(d == 1, 2, 3)).astype(np.uint8)
            if force_dir is not None:
                t["mb_dir"][:] = force_dir
    t["intra"] = intra
    planes = []
    for comp in range(4 if alpha else 3):
        W, H = (cw, ch) if comp in (0, 3) else (cw // 2, ch // 2)
        chroma = comp in (1, 2)
        pix = smooth_scene(rng, W, H)
        res = smooth_scene(rng, W, H, lo=-48.0, hi=48.0, noise=6.0)
        ia_px = np.kron(_mb_to_blocks(intra, mbw, mbh, chroma), np.ones((8, 8), dtype=np.uint8)).astype(bool)
        content = np.where(ia_px, pix, res).astype(np.float32)
        lv = quantise_plane(fdct_plane(content), qscale, intra, mbw, mbh, chroma, qm_intra, qm_non)
        if ptype != PIC_I:
            # uncoded blocks (cbp) and skipped macroblocks carry no coefficients
            unc = rng.random(lv.shape[:2]) < uncoded_frac
            unc |= _mb_to_blocks(skipped, mbw, mbh, chroma)
            unc &= ~_mb_to_blocks(intra, mbw, mbh, chroma).astype(bool)
            lv[unc] = 0
        planes.append(np.ascontiguousarray(blocks_to_plane(lv)))
    t["coef_y"], t["coef_cb"], t["coef_cr"] = planes[:3]
    if alpha:
        t["coef_a"] = planes[3]
    return t


def gop_ibbp(n=12):
    """Closed GOP in CODED order: I B B P B B P ...  Entries (type, display_index,
    fwd_display, bwd_display).  Display order is B0 B1 I2 B3 B4 P5 ...; the two
    leading B pictures use backward prediction only (closed_gop)."""
    out = [(PIC_I, 2, None, None), (PIC_B, 0, None, 2), (PIC_B, 1, None, 2)]
    prev = 2
    d = 5
    while len(out) < n:
        out.append((PIC_P, d, prev, None))
        for b in (d - 2, d - 1):
            if len(out) < n:
                out.append((PIC_B, b, prev, d))
        prev = d
        d += 3
    return out


def gop_ippp(n=16):
    return [(PIC_I, 0, None, None)] + [(PIC_P, i, i - 1, None) for i in range(1, n)]


def dependency_levels(gop):
    """Group a coded-order GOP into launches of mutually independent pictures."""
    level = {}
    for ptype, disp, f, b in gop:
        lv = 0
        if f is not None:
            lv = max(lv, level[f] + 1)
        if b is not None:
            lv = max(lv, level[b] + 1)
        level[disp] = lv
    n = max(level.values()) + 1
    return [[e for e in gop if level[e[1]] == k] for k in range(n)]
