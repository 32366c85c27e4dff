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


def decode_picture(ptype, cw, ch, coef_y, coef_cb, coef_cr, qscale, intra, repadd=None,
                   mb_dir=None, mv_fwd=None, mv_bwd=None, qm=None, pm=None,
                   ref_fwd=None, ref_bwd=None):
    """Returns the decoded picture as one uint8 array [Y | Cb | Cr] of cw*ch*3/2 bytes."""
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


def ycbcr_to_rgba(y, cb, cr, coded_w, frame_w, frame_h, mode="cpu"):
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
