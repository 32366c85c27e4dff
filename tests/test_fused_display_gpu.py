"""GPU: the fused display conversion (leon_picture.rgba_out, ABI 2) -- reconstruction and the
CPU-twin RGBA conversion in ONE kernel -- against the oracle: planes where they are written, RGBA
bit for bit (= jsv.prototype.YCbCrToRGBA, player/easybits.player.js:2674-2785, the parity target of
decision D10), and the planes of no_planes pictures left untouched."""
import numpy as np
import pytest

from helpers import planes_flat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import leon_ctypes
    leon_ctypes.load()
    return leon_ctypes


@pytest.fixture(scope="module")
def O():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="module")
def S():
    import synth
    return synth


def _gop(S, rng, cw, ch, gop, **kw):
    out = {}
    for ptype, disp, f, b in gop:
        force = 2 if (ptype == S.PIC_B and f is None) else None
        out[disp] = S.make_picture(rng, cw, ch, ptype, force_dir=force, **kw)
    return out


def _decode_oracle(O, cw, ch, gop, tens):
    outs = {}
    for ptype, disp, f, b in gop:
        t = tens[disp]
        fwd = f if f is not None else b
        outs[disp] = O.decode_picture(ptype, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                                      repadd=t.get("repadd"), mb_dir=t.get("mb_dir"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                                      ref_fwd=None if fwd is None else outs[fwd], ref_bwd=None if b is None else outs[b])
    return outs


@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
@pytest.mark.parametrize("size", [(352, 240, 352, 240), (96, 64, 96, 64), (208, 112, 200, 104), (1920, 1088, 1920, 1080)],
                         ids=lambda s: "%dx%d" % (s[2], s[3]))
def test_fused_gop_equals_oracle(L, O, S, size, sparse):
    import torch
    import leon_vlc_ctypes as V
    cw, ch, fw, fh = size
    gop = S.gop_ibbp(9 if cw < 1000 else 6)
    rng = np.random.default_rng(cw * 7 + sparse)
    tens = _gop(S, rng, cw, ch, gop, in_picture=cw < 1000)        # small sizes: vectors may leave the picture
    exp = _decode_oracle(O, cw, ch, gop, tens)
    n = len(gop)
    dec = L.Decoder(cw, ch, fw, fh, n_slots=n + 1)
    try:
        rgba = torch.zeros((n, fh, fw, 4), dtype=torch.uint8, device="cuda")
        marker = np.full(cw * ch * 3 // 2, 0xA5, dtype=np.uint8)
        keep = []
        for ptype, disp, f, b in gop:
            t = tens[disp]
            is_b = ptype == S.PIC_B
            if is_b:      # its planes must stay as they are
                dec.write_planes(disp, *O.split_planes(marker, cw, ch))
            fwd = f if f is not None else b
            kw = dict(repadd=t.get("repadd"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"), mb_dir=t.get("mb_dir"),
                      ref_fwd_slot=-1 if fwd is None else fwd, ref_bwd_slot=-1 if b is None else b, keep=keep,
                      rgba_out=rgba[disp].data_ptr(), no_planes=is_b)
            if sparse:
                go, en = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], cw, ch)
                dec.submit_sparse([L.make_sparse_picture(ptype, disp, go, en, len(en), t["qscale"], t["intra"], **kw)], L.MEM_HOST)
            else:
                dec.submit_picture(L.make_picture(ptype, disp, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], **kw))
        dec.sync()
        got = rgba.cpu().numpy()
        for ptype, disp, f, b in gop:
            y, cb, cr = O.split_planes(exp[disp], cw, ch)
            want = O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "cpu")
            bad = np.nonzero(got[disp] != want)
            assert bad[0].size == 0, "RGBA of picture %d (type %d): %d bytes differ, first at row %d col %d" % (
                disp, ptype, bad[0].size, bad[0][0], bad[1][0])
            planes = planes_flat(*dec.read_planes(disp))
            if ptype == S.PIC_B:
                assert (planes == 0xA5).all(), "a no_planes picture wrote its planes"
            else:
                assert np.array_equal(planes, exp[disp]), "planes of picture %d" % disp
        # the unfused path on the same decoder gives the same frames
        for ptype, disp, f, b in gop:
            if ptype != S.PIC_B:
                assert np.array_equal(dec.convert_rgba(disp), got[disp])
    finally:
        dec.close()


def test_fused_conversion_of_every_y_cb_cr(L, O, S):
    """all 2^24 (Y, Cb, Cr) through the fused kernel's table conversion (leon_rgba_lut.h): a 4096 x 4096 P picture
    without residual and with zero vectors hands the planes written into its reference slot to stage 5; the 2x2
    quads run through all (Cb, Cr), the samples of the 64 quads that share a pair through all Y"""
    import torch
    import leon_vlc_ctypes as V
    n = 4096
    q = np.arange(n // 2)
    qy, qx = np.meshgrid(q, q, indexing="ij")
    cb = (qx & 255).astype(np.uint8)
    cr = (qy & 255).astype(np.uint8)
    k = ((qx >> 8) + 8 * (qy >> 8)).astype(np.int64)                       # 0..63: which of the 64 quads of its (Cb, Cr)
    y = np.empty((n, n), np.uint8)
    for dy in range(2):
        for dx in range(2):
            y[dy::2, dx::2] = 4 * k + 2 * dy + dx
    seen = np.zeros((256, 256, 256), bool)
    seen[y.reshape(n // 2, 2, n // 2, 2), cb[:, None, :, None], cr[:, None, :, None]] = True
    assert seen.all()
    mbs = (n // 16) ** 2
    zeros = np.zeros((n, n), np.int16)
    go, en = V.sparsify(zeros, zeros[:n // 2, :n // 2], zeros[:n // 2, :n // 2], n, n)
    assert len(en) == 0
    dec = L.Decoder(n, n, n_slots=2)
    try:
        dec.write_planes(0, y, cb, cr)
        rgba = torch.zeros((n, n, 4), dtype=torch.uint8, device="cuda")
        keep = []
        pic = L.make_sparse_picture(S.PIC_P, 1, go, np.zeros(1, np.uint32), 0, np.ones(mbs, np.uint8), np.zeros(mbs, np.uint8),
                                    repadd=np.zeros(mbs, np.uint8), mv_fwd=np.zeros((mbs, 2), np.int16), ref_fwd_slot=0, keep=keep,
                                    rgba_out=rgba.data_ptr())
        dec.submit_sparse([pic], L.MEM_HOST)
        dec.sync()
        assert np.array_equal(planes_flat(*dec.read_planes(1)), planes_flat(y, cb, cr))
        want = O.ycbcr_to_rgba(y, cb, cr, n, n, n, "cpu").reshape(n, n, 4)
        got = rgba.cpu().numpy()
        bad = np.argwhere(got != want)
        assert bad.size == 0, "%d bytes differ, first at %s: Y %d Cb %d Cr %d -> %s, oracle %s" % (
            len(bad), bad[0], y[bad[0][0], bad[0][1]], cb[bad[0][0] // 2, bad[0][1] // 2], cr[bad[0][0] // 2, bad[0][1] // 2],
            got[bad[0][0], bad[0][1]], want[bad[0][0], bad[0][1]])
    finally:
        dec.close()


def test_fused_batch_mixed_with_plain_pictures(L, O, S):
    """one device batch holding fused and plain pictures of the same type: two launch classes"""
    import torch
    cw, ch = 96, 64
    rng = np.random.default_rng(77)
    tens = [S.make_picture(rng, cw, ch, S.PIC_I) for _ in range(4)]
    dec = L.Decoder(cw, ch, n_slots=4)
    try:
        rgba = torch.zeros((4, ch, cw, 4), dtype=torch.uint8, device="cuda")
        keep, pics = [], []
        dev = lambda a: (keep.append(torch.from_numpy(np.ascontiguousarray(a)).cuda()), keep[-1].data_ptr())[1]
        for k, t in enumerate(tens):
            pics.append(L.make_picture(S.PIC_I, k, dev(t["coef_y"]), dev(t["coef_cb"]), dev(t["coef_cr"]), dev(t["qscale"]), dev(t["intra"]),
                                       device=True, rgba_out=rgba[k].data_ptr() if k & 1 else None))
        dec.submit_batch(pics, L.MEM_DEVICE)
        dec.sync()
        for k, t in enumerate(tens):
            exp = O.decode_picture(S.PIC_I, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"])
            assert np.array_equal(planes_flat(*dec.read_planes(k)), exp)
            if k & 1:
                y, cb, cr = O.split_planes(exp, cw, ch)
                assert np.array_equal(rgba[k].cpu().numpy(), O.ycbcr_to_rgba(y, cb, cr, cw, cw, ch, "cpu"))
            else:
                assert int(rgba[k].sum()) == 0
    finally:
        dec.close()


def test_fused_display_argument_errors(L, S):
    import torch
    rng = np.random.default_rng(1)
    t = S.make_picture(rng, 96, 64, S.PIC_I)
    dec = L.Decoder(96, 64, 90, 60, n_slots=2)          # frame width not a multiple of 8
    try:
        buf = torch.zeros(96 * 64 * 4 + 64, dtype=torch.uint8, device="cuda")
        with pytest.raises(L.LeonError) as e:
            dec.submit_picture(L.make_picture(S.PIC_I, 0, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], rgba_out=buf.data_ptr()))
        assert "frame_width % 8" in str(e.value)
    finally:
        dec.close()
    dec = L.Decoder(96, 64, n_slots=2)
    try:
        with pytest.raises(L.LeonError):               # misaligned destination
            dec.submit_picture(L.make_picture(S.PIC_I, 0, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], rgba_out=buf.data_ptr() + 4))
        with pytest.raises(L.LeonError):               # no_planes without a destination
            dec.submit_picture(L.make_picture(S.PIC_I, 0, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], no_planes=True))
    finally:
        dec.close()
