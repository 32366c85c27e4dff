"""ctypes loader for the CPU oracle (oracle/libleon_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libleon_oracle.so")

PIC_I, PIC_P, PIC_B = 1, 2, 3


def build(force=False):
    src = os.path.join(_HERE, "leon_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.lo_handoff_store.restype = C.c_int32
        _lib.lo_handoff_store.argtypes = [C.c_int32]
    return _lib


def _p(a, dt):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dt)
    return a, a.ctypes.data_as(C.c_void_p)


def default_qm():
    l = lib()
    intra = np.frombuffer((C.c_uint8 * 64).in_dll(l, "LO_DEFAULT_INTRA_QUANT"), dtype=np.uint8).copy()
    non = np.frombuffer((C.c_uint8 * 64).in_dll(l, "LO_DEFAULT_NON_INTRA_QUANT"), dtype=np.uint8).copy()
    return np.concatenate([intra, non])


def premultiplier():
    return np.frombuffer((C.c_uint8 * 64).in_dll(lib(), "LO_PREMULTIPLIER"), dtype=np.uint8).copy()


def butterfly8(x):
    xi = np.ascontiguousarray(x, dtype=np.int32)
    o = np.zeros(8, dtype=np.int32)
    lib().lo_butterfly8(xi.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    return o


def pass1_plane(coef, W, H, is_chroma, qscale, intra, mbw, qm, pm):
    coef = np.ascontiguousarray(coef, dtype=np.int16)
    qs = np.ascontiguousarray(qscale, dtype=np.uint8)
    ia = np.ascontiguousarray(intra, dtype=np.uint8)
    qm = np.ascontiguousarray(qm, dtype=np.uint8)
    pm = np.ascontiguousarray(pm, dtype=np.uint8)
    out = np.zeros(W * H, dtype=np.int16)
    lib().lo_pass1_plane(coef.ctypes.data_as(C.c_void_p), W, H, int(is_chroma),
                         qs.ctypes.data_as(C.c_void_p), ia.ctypes.data_as(C.c_void_p), mbw,
                         qm.ctypes.data_as(C.c_void_p), pm.ctypes.data_as(C.c_void_p),
                         out.ctypes.data_as(C.c_void_p))
    return out.reshape(H, W)


def predict_plane(ref, W, H, is_chroma, mv, mbw):
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    mv = np.ascontiguousarray(mv, dtype=np.int16)
    out = np.zeros(W * H, dtype=np.uint8)
    lib().lo_predict_plane(ref.ctypes.data_as(C.c_void_p), W, H, int(is_chroma),
                           mv.ctypes.data_as(C.c_void_p), mbw, out.ctypes.data_as(C.c_void_p))
    return out.reshape(H, W)


def decode_alpha_plane(ptype, cw, ch, coef_a, qscale, intra, repadd=None, mb_dir=None, mv_fwd=None, mv_bwd=None,
                       qm=None, pm=None, ref_fwd_a=None, ref_bwd_a=None):
    """The fourth component of a yuva picture (container flag `a`, decoders/jsv.js:256-259; 4-plane
    output ring :59-73): a luma-sized plane reconstructed exactly like luma -- same maps, same (luma)
    vectors, same matrices -- from its own coefficient plane and the references' alpha planes.  The
    reference allocates the plane and never decodes it (IDCT_GL loops over three components,
    :1223), so this is the repo's definition, composed from the pinned per-plane functions."""
    qm = default_qm() if qm is None else np.ascontiguousarray(qm, dtype=np.uint8)
    pm = premultiplier() if pm is None else np.ascontiguousarray(pm, dtype=np.uint8)
    mbw = cw // 16
    scratch = pass1_plane(coef_a, cw, ch, False, qscale, intra, mbw, qm, pm)
    out = np.zeros(cw * ch, dtype=np.uint8)
    p = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)
    sc = np.ascontiguousarray(scratch, dtype=np.int16)
    if ptype == 1:
        lib().lo_pass2_intra_plane(sc.ctypes.data_as(C.c_void_p), cw, ch, out.ctypes.data_as(C.c_void_p))
        return out
    rep, mvf = p(repadd, np.uint8), p(mv_fwd, np.int16)
    rf = p(ref_fwd_a, np.uint8)
    if ptype == 2:
        lib().lo_pass2_inter_plane(sc.ctypes.data_as(C.c_void_p), cw, ch, 0, rep.ctypes.data_as(C.c_void_p),
                                   mvf.ctypes.data_as(C.c_void_p), mbw, rf.ctypes.data_as(C.c_void_p),
                                   out.ctypes.data_as(C.c_void_p))
        return out
    md, mvb, rb = p(mb_dir, np.uint8), p(mv_bwd, np.int16), p(ref_bwd_a, np.uint8)
    lib().lo_pass2_bidir_plane(sc.ctypes.data_as(C.c_void_p), cw, ch, 0, rep.ctypes.data_as(C.c_void_p),
                               md.ctypes.data_as(C.c_void_p), mvf.ctypes.data_as(C.c_void_p), mvb.ctypes.data_as(C.c_void_p), mbw,
                               rf.ctypes.data_as(C.c_void_p), rb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def decode_picture(ptype, cw, ch, coef_y, coef_cb, coef_cr, qscale, intra, repadd=None,
                   mb_dir=None, mv_fwd=None, mv_bwd=None, qm=None, pm=None,
                   ref_fwd=None, ref_bwd=None, coef_a=None):
    """Returns the decoded picture as one uint8 array [Y | Cb | Cr] of cw*ch*3/2 bytes -- with coef_a
    (yuva) [Y | Cb | Cr | A] of cw*ch*5/2 bytes, the references in the same layout."""
    if coef_a is not None:
        n3 = cw * ch * 3 // 2
        base = decode_picture(ptype, cw, ch, coef_y, coef_cb, coef_cr, qscale, intra, repadd=repadd, mb_dir=mb_dir,
                              mv_fwd=mv_fwd, mv_bwd=mv_bwd, qm=qm, pm=pm,
                              ref_fwd=None if ref_fwd is None else ref_fwd[:n3], ref_bwd=None if ref_bwd is None else ref_bwd[:n3])
        a = decode_alpha_plane(ptype, cw, ch, coef_a, qscale, intra, repadd=repadd, mb_dir=mb_dir, mv_fwd=mv_fwd, mv_bwd=mv_bwd,
                               qm=qm, pm=pm, ref_fwd_a=None if ref_fwd is None else ref_fwd[n3:],
                               ref_bwd_a=None if ref_bwd is None else ref_bwd[n3:])
        return np.concatenate([base, a])
    keep = []

    def ptr(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)
    if qm is None:
        qm = default_qm()
    if pm is None:
        pm = premultiplier()
    out = np.zeros(cw * ch * 3 // 2, dtype=np.uint8)
    lib().lo_decode_picture(int(ptype), cw, ch, ptr(coef_y, np.int16), ptr(coef_cb, np.int16),
                            ptr(coef_cr, np.int16), ptr(qscale, np.uint8), ptr(intra, np.uint8),
                            ptr(repadd, np.uint8), ptr(mb_dir, np.uint8), ptr(mv_fwd, np.int16),
                            ptr(mv_bwd, np.int16), ptr(qm, np.uint8), ptr(pm, np.uint8),
                            ptr(ref_fwd, np.uint8), ptr(ref_bwd, np.uint8),
                            out.ctypes.data_as(C.c_void_p))
    return out


def split_planes(buf, cw, ch):
    n = cw * ch
    return (buf[:n].reshape(ch, cw), buf[n:n + n // 4].reshape(ch // 2, cw // 2),
            buf[n + n // 4:n + n // 2].reshape(ch // 2, cw // 2))


def ycbcr_to_rgba(y, cb, cr, coded_w, frame_w, frame_h, mode="cpu", a=None):
    """a: the alpha plane of a yuva picture -- the A byte of every pixel the conversion writes is the
    plane's sample instead of 255 (pixels the CPU twin's quad loop never reaches keep fillArray's 255)"""
    if a is not None:
        out = ycbcr_to_rgba(y, cb, cr, coded_w, frame_w, frame_h, mode)
        ap = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, coded_w)[:frame_h, :frame_w]
        if mode == "cpu":       # the quad loop covers (frame_w >> 1) x (frame_h >> 1) quads (even sizes: everything)
            w2, h2 = (frame_w >> 1) * 2, (frame_h >> 1) * 2
            if frame_w & 1:
                raise ValueError("alpha with an odd frame width is not defined (the CPU twin's index drift)")
            out[:h2, :w2, 3] = ap[:h2, :w2]
        else:
            out[:, :, 3] = ap
        return out
    y = np.ascontiguousarray(y, dtype=np.uint8)
    cb = np.ascontiguousarray(cb, dtype=np.uint8)
    cr = np.ascontiguousarray(cr, dtype=np.uint8)
    out = np.zeros(frame_w * frame_h * 4, dtype=np.uint8)
    fn = lib().lo_ycbcr_to_rgba_cpu if mode == "cpu" else lib().lo_ycbcr_to_rgba_gl
    fn(y.ctypes.data_as(C.c_void_p), cb.ctypes.data_as(C.c_void_p), cr.ctypes.data_as(C.c_void_p),
       coded_w, frame_w, frame_h, out.ctypes.data_as(C.c_void_p))
    return out.reshape(frame_h, frame_w, 4)


def dump_gop(dirname, cw, ch, fw, fh, gop, pictures, qm=None):
    """Write one GOP's boundary tensors for oracle/js_baseline.js: manifest.json + raw arrays.
    gop: [(type, display, fwd, bwd)] in coded order; pictures: {display: tensors dict}."""
    import json
    import os
    os.makedirs(dirname, exist_ok=True)
    man = {"cw": cw, "ch": ch, "fw": fw, "fh": fh, "gop": [],
           "qm": [int(v) for v in (default_qm() if qm is None else qm)]}
    dts = {"coef_y": "<i2", "coef_cb": "<i2", "coef_cr": "<i2", "qscale": "u1", "intra": "u1", "repadd": "u1",
           "mb_dir": "u1", "mv_fwd": "<i2", "mv_bwd": "<i2"}
    for ptype, disp, f, b in gop:
        t = pictures[disp]
        files = {}
        for k, dt in dts.items():
            if t.get(k) is not None:
                name = "p%d_%s.bin" % (disp, k)
                np.ascontiguousarray(t[k]).astype(dt).tofile(os.path.join(dirname, name))
                files[k] = name
            else:
                files[k] = None
        man["gop"].append({"type": int(ptype), "disp": int(disp), "fwd": None if f is None else int(f),
                           "bwd": None if b is None else int(b), "files": files})
    with open(os.path.join(dirname, "manifest.json"), "w") as fh_:
        json.dump(man, fh_)
