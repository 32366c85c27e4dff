"""Literal per-fragment emulation of the reference's composed GLSL programs.

TEST INFRASTRUCTURE (part of the oracle tooling, never used by the product).

This is a second, independent restatement of /root/reference
decoders/shaders/mpeg1video.js:18-29 (integer flavour, as composed by
decoders/jsv.js:2461-2464) and of the GL driver calls in
decoders/jsv.js:1177-1336 (texture formats, viewports, FBO sizes) and the
Y-flipping vertex shader player/parts/end.js:158-166.  Unlike oracle/leon_oracle.c,
which works on image-space planes with integer arithmetic, this file emulates the
GL machine: RGBA8 / LUMINANCE(_ALPHA) textures, NEAREST + CLAMP_TO_EDGE sampling
at normalised float32 coordinates, one fragment per output texel with its varying
`_S`, float32 arithmetic exactly where the shader text uses `float`, int32 where
it uses `int` (C division), UNORM8 render-target stores.  Every statement of the
shader text has a line here in the same order, vectorised over all fragments.

It exists to pin the oracle: tests/test_oracle_vs_literal.py requires the two to
agree bit-for-bit, and tools/make_golden.py freezes its outputs as fixtures.

Model decisions: D1 int '/' truncates; D2 float = binary32
round-to-nearest-even, no contraction; D5 UNORM8 store = round(255*clamp(x));
D8 a lookup that lands exactly on a texel edge reads the texel to its right/up
(`_edge_eps`), i.e. "this block's own macroblock".
"""
import numpy as np

F = np.float32
_edge_eps = 1e-4


def f32(x):
    return np.asarray(x, dtype=np.float32)


class Tex:
    """A GL texture: uint8 array [h][w][4] already expanded to RGBA semantics."""

    def __init__(self, rgba):
        self.a = np.ascontiguousarray(rgba, dtype=np.uint8)
        self.h, self.w = self.a.shape[:2]

    @staticmethod
    def luminance(arr2d):
        a = np.asarray(arr2d, dtype=np.uint8)
        return Tex(np.stack([a, a, a, np.full_like(a, 255)], axis=-1))

    @staticmethod
    def luminance_alpha(arr3d):
        a = np.asarray(arr3d, dtype=np.uint8)  # [h][w][2] = (L, A)
        return Tex(np.stack([a[..., 0], a[..., 0], a[..., 0], a[..., 1]], axis=-1))

    @staticmethod
    def rgba(arr3d):
        return Tex(arr3d)

    def fetch(self, u, v, edge=False):
        """texture2D(sampler, vec2(u, v)) with NEAREST/CLAMP_TO_EDGE -> float32 RGBA in [0,1]."""
        eps = _edge_eps if edge else 0.0
        x = np.floor(np.asarray(u, dtype=np.float64) * self.w + eps).astype(np.int64)
        y = np.floor(np.asarray(v, dtype=np.float64) * self.h + eps).astype(np.int64)
        x = np.clip(x, 0, self.w - 1)
        y = np.clip(y, 0, self.h - 1)
        x, y = np.broadcast_arrays(x, y)
        return (self.a[y, x].astype(np.float32) / F(255.0)).astype(np.float32)


def _frag_coords(vw, vh):
    """Varying _S for a full-viewport quad through SHADER_VERTEX_IDENTITY (end.js:158-166).
    Returns (Sx, Sy) of shape [vh][vw], row j = window row j (0 = bottom = texture row 0)."""
    i = np.arange(vw, dtype=np.float64)
    j = np.arange(vh, dtype=np.float64)
    sx = ((i + 0.5) / vw).astype(np.float32)
    sy = (1.0 - (j + 0.5) / vh).astype(np.float32)   # gl_Position.y = -(2v-1)
    return np.broadcast_to(sx[None, :], (vh, vw)), np.broadcast_to(sy[:, None], (vh, vw))


def _unorm8(c):
    c = np.clip(np.asarray(c, dtype=np.float32), F(0), F(1))
    return np.rint(c.astype(np.float64) * 255.0).astype(np.uint8)


def _E(v0, v1):
    """CONV_INT _E(vec2): two bytes -> signed int16 as float (mpeg1video.js:18)."""
    lo = (v0 * F(255.0) + v1 * F(255.0) * F(256.0)).astype(np.float32)
    hi = ((v0 * F(255.0) - F(256.0)) + (v1 * F(255.0) - F(255.0)) * F(256.0)).astype(np.float32)
    return np.where(v1 < F(0.5), lo, hi).astype(np.float32)


def _B(aw):
    """CONV_INT _B(float): signed -> two bytes /255 (mpeg1video.js:18)."""
    aw = np.where(aw < 0, aw + F(65536.0), aw).astype(np.float32)
    al = np.floor(aw / F(256.0)).astype(np.float32)
    ao = (aw - al * F(256.0)).astype(np.float32)
    return (ao / F(255.0)).astype(np.float32), (al / F(255.0)).astype(np.float32)


def _idiv(a, b):
    """GLSL int '/' under D1: truncation toward zero."""
    a = np.asarray(a, dtype=np.int64)
    q = np.abs(a) // b
    return (np.sign(a) * q).astype(np.int64)


def _butterfly(X):
    """COL_INT_5 / ROWSCOM_INT4 (mpeg1video.js:23, :26); X = list of 8 int64 arrays."""
    b1 = X[4]
    b3 = X[2] + X[6]
    b4 = X[5] - X[3]
    tmp1 = X[1] + X[7]
    tmp2 = X[3] + X[5]
    b6 = X[1] - X[7]
    b7 = tmp1 + tmp2
    m0 = X[0]
    x4 = _idiv(b6 * 473 - b4 * 196 + 128, 256) - b7
    x0 = x4 - _idiv((tmp1 - tmp2) * 362 + 128, 256)
    x1 = m0 - b1
    x2 = _idiv((X[2] - X[6]) * 362 + 128, 256) - b3
    x3 = m0 + b1
    y3 = x1 + x2
    y4 = x3 + b3
    y5 = x1 - x2
    y6 = x3 - b3
    y7 = -x0 - _idiv(b4 * 473 + b6 * 196 + 128, 256)
    return [b7 + y4, x4 + y3, y5 - x0, y6 - y7, y6 + y7, x0 + y5, y3 - x4, y4 - b7]


def _mod(x, y):
    return (x - y * np.floor(x / y)).astype(np.float32)


def pass1_columns(coef_i16, qscale, intra, qm128, pm64, mbw, mbh):
    """Program idct_columns drawn over one component (jsv.js:1223-1268).
    coef_i16: [H][W] int16.  Returns the idct_1d texture as uint8 [H][W/2][4]
    (texture row 0 = bottom window row)."""
    H, W = coef_i16.shape
    le = coef_i16.astype('<i2').view(np.uint8).reshape(H, W, 2)
    G = Tex.luminance_alpha(le)                               # unit 0, :1243
    f_ = Tex.luminance(np.asarray(qm128, np.uint8).reshape(16, 8))   # unit 1, :139-144
    t_ = Tex.luminance(np.asarray(qscale, np.uint8).reshape(mbh, mbw))  # unit 2, :1206
    b_ = Tex.luminance(np.asarray(pm64, np.uint8).reshape(8, 8))       # unit 3, :149-150
    w_ = Tex.luminance(np.asarray(intra, np.uint8).reshape(mbh, mbw))  # unit 4, :1217
    ae, ad = W, H                                             # uniforms :1255-1256
    vw, vh = W // 2, H                                        # viewport :1260
    Sx, Sy = _frag_coords(vw, vh)

    y_ = F(0.4)
    h_ = F(1.0) / (F(ae) / F(8.0))
    d_ = F(1.0) / (F(ad) / F(8.0))
    o_ = F(1.0) / F(ad)
    k_ = F(1.0) / F(ae)
    g_ = F(2.0) * k_
    e_ = o_
    Q = np.floor(np.floor(Sx / k_) / F(8.0)).astype(np.float32)
    R = np.floor(np.floor(Sy / o_) / F(8.0)).astype(np.float32)
    A = _mod(np.floor(Sx / g_), F(4.0))
    z = _mod(np.floor(Sy / e_), F(8.0))
    m_ = (Q * F(8.0) * k_ + z * k_ + k_ / F(2.0)).astype(np.float32)
    l_ = (R * F(8.0) * o_ + o_ / F(2.0)).astype(np.float32)
    X = []
    for i in range(8):                                        # COL_INT_2
        t = G.fetch(m_, (l_ + o_ * F(i)).astype(np.float32))
        X.append(_E(t[..., 0], t[..., 3]))
    ag = (w_.fetch((Q + F(0.5)) * h_, (R + F(0.5)) * d_)[..., 0] > 0).astype(np.int32)   # COL_3
    q_ = np.floor(t_.fetch(Q * h_, R * d_, edge=True)[..., 0] * F(255.0) + F(0.5)).astype(np.float32)
    dc = X[0].copy()                                          # COL_INT_21
    for i in range(8):                                        # COL_31 ..
        nz = X[i] != 0                                        # zero: '_U + 1. > last_non_zero' -> continue
        x = (X[i] * F(2.0)).astype(np.float32)
        tqx = (F(0.075) + z * F(0.125)).astype(np.float32)
        tqy = F(0.075) + F(i) * F(0.125)
        j_ = np.where(ag == 0, F(tqy / F(2.0)) + F(0.5), F(tqy / F(2.0))).astype(np.float32)
        O = np.floor(f_.fetch(tqx, j_)[..., 0] * F(255.0) + F(0.5)).astype(np.float32)
        x = np.where(ag == 0, x + np.where(x < 0, F(-1.0), F(1.0)), x).astype(np.float32)   # COL_INT_3
        x = np.floor(((x * q_).astype(np.float32) * O).astype(np.float32) / F(16.0)).astype(np.float32)
        x = np.where(_mod(x, F(2.0)) == 0, x - np.where(x > 0, F(1.0), F(-1.0)), x).astype(np.float32)
        x = np.minimum(x, F(2047.0))
        x = np.maximum(x, F(-2048.0))
        pmv = np.floor(b_.fetch(tqx, np.full_like(tqx, tqy))[..., 0] * F(255.0) + F(0.5)).astype(np.float32)
        x = (x * pmv).astype(np.float32)
        X[i] = np.where(nz, x, X[i]).astype(np.float32)
    X[0] = np.where((z == 0) & (ag == 1), (dc * F(256.0)).astype(np.float32), X[0])    # COL_4/COL_INT_31
    Xi = [np.trunc(x).astype(np.int64) for x in X]            # int(_X[k])
    o = _butterfly(Xi)

    def enc(v):
        return _B(np.floor(v.astype(np.float32) * y_).astype(np.float32))
    sel = [(o[0], o[1]), (o[2], o[3]), (o[4], o[5]), (o[6], o[7])]
    out = np.zeros((vh, vw, 4), dtype=np.float32)
    for a in range(4):
        aj = enc(sel[a][0])
        ah = enc(sel[a][1])
        m = (A == a)
        out[..., 0] = np.where(m, aj[0], out[..., 0])
        out[..., 1] = np.where(m, aj[1], out[..., 1])
        out[..., 2] = np.where(m, ah[0], out[..., 2])
        out[..., 3] = np.where(m, ah[1], out[..., 3])
    return _unorm8(out)            # COL_5 + RGBA8 render target


def _p(ref, Sx, Sy, g_, e_, ax, ay):
    """_p(_ax,_ay): 4 horizontally consecutive reference pixels (mpeg1video.js:24)."""
    u = (np.sign(ax) * np.floor(np.abs(ax) / F(4.0))).astype(np.float32)
    n = _mod(np.abs(ax), F(4.0))
    vy = (F(1.0) - Sy - ay * e_).astype(np.float32)
    Z = ref.fetch((Sx + u * g_).astype(np.float32), vy)
    Hp = ref.fetch((Sx + (u + F(1.0)) * g_).astype(np.float32), vy)
    Hn = ref.fetch((Sx + (u - F(1.0)) * g_).astype(np.float32), vy)
    C = Z.copy()
    pos = ax > 0
    for nn, idx in ((1, [(Z, 1), (Z, 2), (Z, 3), (Hp, 0)]),
                    (2, [(Z, 2), (Z, 3), (Hp, 0), (Hp, 1)]),
                    (3, [(Z, 3), (Hp, 0), (Hp, 1), (Hp, 2)])):
        m = pos & (n == nn)
        for c, (src, k) in enumerate(idx):
            C[..., c] = np.where(m, src[..., k], C[..., c])
    neg = ~pos & (n != 0)
    for nn, idx in ((3, [(Hn, 1), (Hn, 2), (Hn, 3), (Z, 0)]),
                    (2, [(Hn, 2), (Hn, 3), (Z, 0), (Z, 1)]),
                    (1, [(Hn, 3), (Z, 0), (Z, 1), (Z, 2)])):
        m = neg & (n == nn)
        for c, (src, k) in enumerate(idx):
            C[..., c] = np.where(m, src[..., k], C[..., c])
    return C


def _Ftrunc(a):
    return (np.sign(a) * np.floor(np.abs(a))).astype(np.float32)


def pass2_rows(idct1d_rgba, W, H, inter=False, repadd=None, mv_i16=None, prev_rgba=None,
               mv_coef=1.0, mbw=0, mbh=0):
    """Programs idct_rows_intra / idct_rows_inter drawn over one component
    (jsv.js:1305-1334).  idct1d_rgba: uint8 [H][W/2][4] from pass1_columns;
    prev_rgba: previous output texture uint8 [H][W/4][4].  Returns uint8 [H][W/4][4]."""
    G = Tex.rgba(idct1d_rgba)
    ae, ad = W // 2, H                                        # :1323-1324
    vw, vh = W // 4, H                                        # :1327
    Sx, Sy = _frag_coords(vw, vh)
    y_ = F(0.4)
    o_ = F(1.0) / F(ad)
    k_ = F(1.0) / F(ae)
    g_ = F(2.0) * k_
    e_ = o_
    h_ = F(1.0) / (F(ae) / F(4.0))
    d_ = F(1.0) / (F(ad) / F(8.0))
    Q = np.floor(np.floor(Sx / g_) / F(2.0)).astype(np.float32)
    R = np.floor(np.floor(Sy / e_) / F(8.0)).astype(np.float32)
    A = _mod(np.floor(Sx / g_), F(2.0))
    a_ = (F(7.0) - _mod(np.floor(Sy / e_), F(8.0))).astype(np.float32)
    m_ = (Q * F(4.0) * k_ + np.floor(a_ / F(2.0)) * k_ + k_ * F(0.5)).astype(np.float32)
    l_ = ((R + F(0.95)) * F(8.0) * o_).astype(np.float32)
    even = _mod(a_, F(2.0)) == 0
    X = []
    for i in range(8):
        texel = G.fetch(m_, (l_ - o_ * F(i)).astype(np.float32))
        v = np.where(even, _E(texel[..., 0], texel[..., 1]), _E(texel[..., 2], texel[..., 3]))
        X.append(np.trunc((v.astype(np.float32) / y_).astype(np.float32)).astype(np.int64))  # ROWS_INT1/2
    o = _butterfly(X)
    pix = [(_idiv(v + 128, 256).astype(np.float32) / F(255.0)).astype(np.float32) for v in o]
    out = np.zeros((vh, vw, 4), dtype=np.float32)
    for c in range(4):
        out[..., c] = np.where(A == 0, pix[c], pix[4 + c])
    if inter:                                                 # SHADER_FRAGMENT_IDCT_ROWS_INTER_1
        w_ = Tex.luminance(np.asarray(repadd, np.uint8).reshape(mbh, mbw))     # :1282-1284
        mvb = np.asarray(mv_i16, dtype='<i2').view(np.uint8).reshape(mbh, mbw, 4)
        I_ = Tex.rgba(mvb)                                    # :1296-1298
        i_ = Tex.rgba(prev_rgba)                              # :1319-1320
        rep = w_.fetch((Q + F(0.5)) * h_, F(1.0) - (R + F(0.5)) * d_)[..., 0] > F(0.5)
        ar = I_.fetch(Sx, (F(1.0) - e_ * F(0.25) - Sy).astype(np.float32))
        Wv = _E(ar[..., 0], ar[..., 1])
        Vv = _E(ar[..., 2], ar[..., 3])
        if mv_coef == 1.0:
            ax = np.floor(Wv / F(2.0)).astype(np.float32)
            ay = np.floor(Vv / F(2.0)).astype(np.float32)
            odd_h = np.abs(Wv - ax * F(2.0)) > F(0.5)
            odd_v = np.abs(Vv - ay * F(2.0)) > F(0.5)
        else:
            ax = np.floor(_Ftrunc(Wv / F(2.0)) / F(2.0)).astype(np.float32)
            ay = np.floor(_Ftrunc(Vv / F(2.0)) / F(2.0)).astype(np.float32)
            odd_h = _mod(_Ftrunc(Wv / F(2.0)), F(2.0)) != 0
            odd_v = _mod(_Ftrunc(Vv / F(2.0)), F(2.0)) != 0
        ay = (ay * F(-1.0)).astype(np.float32)
        C = _p(i_, Sx, Sy, g_, e_, ax, ay)
        aa = np.ones_like(Sx, dtype=np.float32)
        am = F(0.001953125)
        C = np.where(odd_h[..., None], C + _p(i_, Sx, Sy, g_, e_, ax + F(1.0), ay) + am, C).astype(np.float32)
        aa = np.where(odd_h, aa * F(2.0), aa)
        C = np.where(odd_v[..., None], C + _p(i_, Sx, Sy, g_, e_, ax, ay - F(1.0)) + am, C).astype(np.float32)
        aa = np.where(odd_v, aa * F(2.0), aa)
        C = np.where((odd_h & odd_v)[..., None],
                     C + _p(i_, Sx, Sy, g_, e_, ax + F(1.0), ay - F(1.0)), C).astype(np.float32)
        C = (C / aa[..., None]).astype(np.float32)
        C = np.where(rep[..., None], F(0.0), C).astype(np.float32)
        out = (out + C).astype(np.float32)                    # INTER_INT1: gl_FragColor += _C
    return _unorm8(out)


def tex_to_plane(rgba):
    """Output/reference texture [H][W/4][4] (row 0 = bottom window row) -> image plane [H][W].
    The vertex shader's flip makes texture row r hold image row r."""
    h, w4, _ = rgba.shape
    return rgba.reshape(h, w4 * 4)


def plane_to_tex(plane):
    h, w = plane.shape
    return np.ascontiguousarray(plane, dtype=np.uint8).reshape(h, w // 4, 4)


def decode_picture_literal(ptype, cw, ch, coef, qscale, intra, repadd, mv, qm128, pm64, prev_planes):
    """IDCT_GL for an I (ptype 1) or P (ptype 2) picture through the emulated GL machine.
    coef: (Y,Cb,Cr) int16 planes; prev_planes: (Y,Cb,Cr) uint8 planes or None.
    Returns (Y, Cb, Cr) uint8 image planes."""
    mbw, mbh = cw // 16, ch // 16
    outs = []
    for comp in range(3):
        W, H = (cw, ch) if comp == 0 else (cw // 2, ch // 2)
        s = pass1_columns(np.asarray(coef[comp]).reshape(H, W), qscale, intra, qm128, pm64, mbw, mbh)
        if ptype == 1:
            o = pass2_rows(s, W, H)
        else:
            o = pass2_rows(s, W, H, inter=True, repadd=repadd, mv_i16=mv,
                           prev_rgba=plane_to_tex(np.asarray(prev_planes[comp]).reshape(H, W)),
                           mv_coef=1.0 if comp == 0 else 0.5, mbw=mbw, mbh=mbh)
        outs.append(tex_to_plane(o))
    return outs
