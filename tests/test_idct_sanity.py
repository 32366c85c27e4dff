"""CPU sanity check (not a parity gate): the oracle's dequantiser + two-pass integer
IDCT, driven through the picture interface, against a double-precision inverse DCT of the same
coefficients -- IEEE-1180-style statistics.  The reference's lossy x0.4 / x2.5 int16 hand-off between
the passes costs accuracy, so the bounds are looser than the standard's; what this test pins is that
the restated chain (premultiplier, butterfly constants, /256 scalings, transposes) is an IDCT at all."""
import numpy as np

from oracle import oracle_py as O


def _idct2(F):
    """orthonormal 8x8 inverse DCT, float64; F[..., v, u] (row = vertical frequency)"""
    k = np.arange(8)
    C = np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / 16)        # [x][u]
    a = np.full(8, 0.5)
    a[0] = np.sqrt(0.125)
    B = C * a[None, :]
    return np.einsum("yv,...vu,xu->...yx", B, F, B)


def _dct2(f):
    k = np.arange(8)
    C = np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / 16)
    a = np.full(8, 0.5)
    a[0] = np.sqrt(0.125)
    B = C * a[None, :]
    return np.einsum("yv,...yx,xu->...vu", B, f, B)


def test_two_pass_integer_idct_tracks_a_double_precision_idct():
    O.lib()
    rng = np.random.default_rng(1180)
    cw, ch = 256, 128                               # 512 luma blocks + 256 chroma blocks per picture
    nmb = (cw // 16) * (ch // 16)
    errs = []
    # coefficients of real residual blocks (the pass-1 hand-off is int16: spectra no 8-bit picture can
    # produce wrap there, in the reference as in the oracle -- SURVEY.md D6)
    for lo, hi in ((-5, 5), (-60, 60), (-120, 120)):
        planes, F = [], []
        for W, H in ((cw, ch), (cw // 2, ch // 2), (cw // 2, ch // 2)):
            px = rng.uniform(lo, hi, (H // 8, W // 8, 8, 8))
            px = 0.5 * px + 0.5 * px.mean(axis=(2, 3), keepdims=True)      # some DC, some texture
            lv = np.trunc(_dct2(px) / 16.0).astype(np.int16)
            planes.append(np.ascontiguousarray(lv.transpose(0, 2, 1, 3).reshape(H, W)))
            # non-intra, quantiser_scale 8, flat matrix 16: (2L + sign) * 8 * 16 / 16, then oddified toward zero
            F.append(np.sign(lv) * (16 * np.abs(lv.astype(np.int64)) + 7))
        ref = np.full(cw * ch * 3 // 2, 128, np.uint8)
        out = O.decode_picture(2, cw, ch, planes[0].ravel(), planes[1].ravel(), planes[2].ravel(),
                               np.full(nmb, 8, np.uint8), np.zeros(nmb, np.uint8), repadd=np.zeros(nmb, np.uint8),
                               mv_fwd=np.zeros(2 * nmb, np.int16), ref_fwd=ref)
        y, cb, cr = O.split_planes(out, cw, ch)
        for got, f, (W, H) in zip((y, cb, cr), F, ((cw, ch), (cw // 2, ch // 2), (cw // 2, ch // 2))):
            exact = _idct2(f.astype(np.float64))                                   # [R][Q][y][x]
            want = np.clip(np.rint(exact) + 128, 0, 255)
            g = got.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).astype(np.float64)
            ok = (want > 0) & (want < 255)                                          # ignore clipped samples
            errs.append((g - want)[ok])
    e = np.concatenate(errs)
    peak, mse, mean = np.abs(e).max(), float((e ** 2).mean()), float(e.mean())
    print("two-pass IDCT vs float64: peak %.0f, mse %.3f, mean %.3f over %d samples" % (peak, mse, mean, e.size))
    # measured: peak 6, mse 0.83, mean +0.33.  IEEE 1180 asks an exact-hand-off IDCT for peak 1 / mse 0.02 /
    # mean 0.0015; this chain has an 8-bit premultiplier, truncates at the x0.4 hand-off and again at /256
    # (toward zero: a positive bias around the +128 offset) -- the reference's arithmetic, restated, not tuned
    assert peak <= 8 and mse < 1.5 and abs(mean) < 0.6, (peak, mse, mean)
