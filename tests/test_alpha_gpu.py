"""GPU: yuva -- the fourth, luma-sized component behind the container's `a` flag (decoders/jsv.js:256-259)
and the reference's 4-plane output ring (jsv.js:59-73).  The reference allocates the plane and never
decodes it (IDCT_GL loops over three components, jsv.js:1223; with `a` set its GL path reads an unbound
premultiplier, :1208-1211), so there is nothing of the reference to pin this to: the oracle composes the
A plane from the reference-pinned per-plane functions (pass 1, pass 2, forward/bidirectional MC), and the
HIP path must match it bit for bit -- dense and sparse boundary, planes and RGBA."""
import numpy as np
import pytest

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


@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
@pytest.mark.parametrize("size", [(96, 64, 96, 64), (208, 112, 200, 106), (352, 240, 352, 240)], ids=lambda s: "%dx%d" % (s[2], s[3]))
def test_yuva_gop_equals_oracle(L, O, S, size, sparse):
    import leon_vlc_ctypes as V
    cw, ch, fw, fh = size
    gop = S.gop_ibbp(9)
    rng = np.random.default_rng(cw + 11 * sparse)
    dec = L.Decoder(cw, ch, fw, fh, n_slots=len(gop), alpha=True)
    try:
        outs, keep = {}, []
        for ptype, disp, f, b in gop:
            t = S.make_picture(rng, cw, ch, ptype, alpha=True, in_picture=False, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            fwd = f if f is not None else b
            outs[disp] = O.decode_picture(ptype, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                                          repadd=t.get("repadd"), mb_dir=t.get("mb_dir"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                                          ref_fwd=None if fwd is None else outs[fwd], ref_bwd=None if b is None else outs[b],
                                          coef_a=t["coef_a"])
            kw = dict(repadd=t.get("repadd"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"), mb_dir=t.get("mb_dir"),
                      ref_fwd_slot=-1 if fwd is None else fwd, ref_bwd_slot=-1 if b is None else b, keep=keep)
            if sparse:
                go, en = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], cw, ch, coef_a=t["coef_a"])
                dec.submit_sparse([L.make_sparse_picture(ptype, disp, go, en, len(en), t["qscale"], t["intra"], **kw)], L.MEM_HOST)
            else:
                dec.submit_picture(L.make_picture(ptype, disp, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                                                  coef_a=t["coef_a"], **kw))
        dec.sync()
        n3 = cw * ch * 3 // 2
        for disp, exp in outs.items():
            y, cb, cr = dec.read_planes(disp)
            assert np.array_equal(np.concatenate([y.ravel(), cb.ravel(), cr.ravel()]), exp[:n3]), "Y/Cb/Cr of picture %d" % disp
            a = dec.read_alpha_plane(disp)
            bad = np.nonzero(a.ravel() != exp[n3:])[0]
            assert bad.size == 0, "A plane of picture %d: %d samples differ, first at %d" % (disp, bad.size, bad[0])
            ey, ecb, ecr = O.split_planes(exp[:n3], cw, ch)
            for flavour, mode in ((L.RGB_CPU_TWIN, "cpu"), (L.RGB_GL, "gl")):
                want = O.ycbcr_to_rgba(ey, ecb, ecr, cw, fw, fh, mode, a=exp[n3:])
                assert np.array_equal(dec.convert_rgba(disp, flavour), want), "RGBA (%s) of picture %d" % (mode, disp)
    finally:
        dec.close()


@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
@pytest.mark.parametrize("size", [(96, 64, 96, 64), (208, 112, 200, 106), (352, 240, 352, 240), (1920, 1088, 1920, 1080)],
                         ids=lambda s: "%dx%d" % (s[2], s[3]))
def test_yuva_fused_display_equals_oracle(L, O, S, size, sparse):
    """rgba_out on a yuva decoder: the A part of a task runs before the Y part of the same four macroblocks
    and its samples become the pixels' A bytes (k_recon_display<.., .., true>); B pictures write no planes."""
    import torch
    import leon_vlc_ctypes as V
    cw, ch, fw, fh = size
    gop = S.gop_ibbp(9 if cw < 1000 else 6)
    rng = np.random.default_rng(cw + 5 * sparse)
    dec = L.Decoder(cw, ch, fw, fh, n_slots=len(gop), alpha=True)
    n3, n4 = cw * ch * 3 // 2, cw * ch * 5 // 2
    try:
        rgba = torch.zeros((len(gop), fh, fw, 4), dtype=torch.uint8, device="cuda")
        sentinel = np.full(n4, 77, np.uint8)
        outs, keep = {}, []
        for ptype, disp, f, b in gop:
            t = S.make_picture(rng, cw, ch, ptype, alpha=True, in_picture=False, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            fwd = f if f is not None else b
            outs[disp] = O.decode_picture(ptype, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                                          repadd=t.get("repadd"), mb_dir=t.get("mb_dir"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                                          ref_fwd=None if fwd is None else outs[fwd], ref_bwd=None if b is None else outs[b],
                                          coef_a=t["coef_a"])
            no_planes = ptype == S.PIC_B
            if no_planes:
                y, cb, cr = O.split_planes(sentinel[:n3], cw, ch)
                dec.write_planes(disp, y, cb, cr)
                dec.write_alpha_plane(disp, sentinel[n3:].reshape(ch, cw))
            kw = dict(repadd=t.get("repadd"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"), mb_dir=t.get("mb_dir"),
                      ref_fwd_slot=-1 if fwd is None else fwd, ref_bwd_slot=-1 if b is None else b, keep=keep,
                      rgba_out=rgba[disp].data_ptr(), no_planes=no_planes)
            if sparse:
                go, en = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], cw, ch, coef_a=t["coef_a"])
                dec.submit_sparse([L.make_sparse_picture(ptype, disp, go, en, len(en), t["qscale"], t["intra"], **kw)], L.MEM_HOST)
            else:
                dec.submit_picture(L.make_picture(ptype, disp, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                                                  coef_a=t["coef_a"], **kw))
        dec.sync()
        got = rgba.cpu().numpy()
        for ptype, disp, f, b in gop:
            exp = outs[disp]
            ey, ecb, ecr = O.split_planes(exp[:n3], cw, ch)
            want = O.ycbcr_to_rgba(ey, ecb, ecr, cw, fw, fh, "cpu", a=exp[n3:])
            bad = np.argwhere(got[disp].reshape(fh, fw, 4) != want.reshape(fh, fw, 4))
            assert bad.size == 0, "RGBA of picture %d: %d bytes differ, first at %s" % (disp, len(bad), bad[0])
            y, cb, cr = dec.read_planes(disp)
            have = np.concatenate([y.ravel(), cb.ravel(), cr.ravel(), dec.read_alpha_plane(disp).ravel()])
            assert np.array_equal(have, sentinel if ptype == S.PIC_B else exp), "planes of picture %d" % disp
    finally:
        dec.close()


def test_yuva_argument_errors(L, S):
    rng = np.random.default_rng(2)
    t = S.make_picture(rng, 96, 64, S.PIC_I, alpha=True)
    with pytest.raises(L.LeonError):
        L.Decoder(96, 64, 95, 64, alpha=True)          # odd frame width
    dec = L.Decoder(96, 64, alpha=True)
    try:
        with pytest.raises(L.LeonError) as e:            # no A coefficients
            dec.submit_picture(L.make_picture(S.PIC_I, 0, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"]))
        assert "coef_a" in str(e.value)
    finally:
        dec.close()
    plain = L.Decoder(96, 64)
    try:
        with pytest.raises(L.LeonError):
            plain.read_alpha_plane(0)
    finally:
        plain.close()
