"""GPU parity of the SPARSE boundary (leon_submit_sparse, include/leon.h): group lists in the
format of include/leon_vlc.h must reconstruct exactly what the dense planes do -- against the CPU
oracle, bit for bit -- for synthetic pictures and for whole streams parsed by the native front end
(stream bytes -> libleon_vlc -> libleon_hip -> planes)."""
import os

import numpy as np
import pytest

from helpers import ROOT, oracle_decode_sequence, hip_submit_sparse, planes_flat

pytestmark = pytest.mark.gpu
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")


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


def _chain(S, rng, cw, ch, gop, **kw):
    pics = []
    for ptype, disp, f, b in gop:
        force = 2 if (ptype == S.PIC_B and f is None) else None
        t = S.make_picture(rng, cw, ch, ptype, force_dir=force, **kw)
        t["slot"] = disp
        t["ref_fwd"] = f if f is not None else (b if ptype == S.PIC_B else None)
        t["ref_bwd"] = b
        pics.append(t)
    return pics


def _run(L, O, cw, ch, pics, qm=None):
    dec = L.Decoder(cw, ch, n_slots=16)
    try:
        if qm is not None:
            dec.set_quant_matrices(qm[:64], qm[64:])
        keep = []
        for t in pics:
            hip_submit_sparse(L, dec, t, keep, cw, ch)
        dec.sync()
        exp = oracle_decode_sequence(O, cw, ch, pics, {}, qm=qm)
        for t in pics:
            got = planes_flat(*dec.read_planes(t["slot"]))
            bad = np.nonzero(got != exp[t["slot"]])[0]
            assert bad.size == 0, "picture type %d slot %d: %d samples differ, first at %d" % (t["type"], t["slot"], bad.size, bad[0])
    finally:
        dec.close()


@pytest.mark.parametrize("size", [(352, 240), (96, 64), (176, 144)])
def test_ibbp_gop_sparse_equals_oracle(L, O, S, size):
    rng = np.random.default_rng(size[0] * 7 + 1)
    _run(L, O, size[0], size[1], _chain(S, rng, size[0], size[1], S.gop_ibbp(12)))


def test_vectors_leave_picture_sparse(L, O, S):
    rng = np.random.default_rng(12)
    _run(L, O, 176, 144, _chain(S, rng, 176, 144, S.gop_ibbp(6), in_picture=False))


def test_groups_longer_than_one_wave_of_entries(L, O):
    """Garbage levels everywhere: 512 entries per group, so the in-kernel list loop runs 8 times."""
    cw, ch = 160, 96
    rng = np.random.default_rng(3)
    nmb = (cw // 16) * (ch // 16)
    pics = []
    for i, amp in enumerate((255, 2047, 32767)):
        t = {"type": 1, "slot": i, "ref_fwd": None, "ref_bwd": None,
             "coef_y": rng.integers(-amp, amp + 1, cw * ch).astype(np.int16),
             "coef_cb": rng.integers(-amp, amp + 1, cw * ch // 4).astype(np.int16),
             "coef_cr": rng.integers(-amp, amp + 1, cw * ch // 4).astype(np.int16),
             "qscale": rng.integers(1, 32, nmb).astype(np.uint8),
             "intra": np.where(rng.random(nmb) < 0.5, 255, 0).astype(np.uint8)}
        pics.append(t)
    _run(L, O, cw, ch, pics)


def test_empty_lists_and_single_entries(L, O, S):
    cw, ch = 64, 48
    nmb = (cw // 16) * (ch // 16)
    z = lambda n: np.zeros(n, np.int16)
    empty = {"type": 1, "slot": 0, "ref_fwd": None, "ref_bwd": None, "coef_y": z(cw * ch), "coef_cb": z(cw * ch // 4),
             "coef_cr": z(cw * ch // 4), "qscale": np.full(nmb, 8, np.uint8), "intra": np.full(nmb, 255, np.uint8)}
    one = dict(empty, slot=1, coef_y=z(cw * ch), coef_cr=z(cw * ch // 4))
    one["coef_y"][cw * 9 + 17] = -3            # one AC coefficient in one block
    one["coef_cr"][0] = 200                    # one DC in the first chroma block
    _run(L, O, cw, ch, [empty, one])


def test_malformed_lists_are_rejected_or_harmless(L):
    import leon_vlc_ctypes as V
    cw, ch = 64, 48
    nmb = 12
    dec = L.Decoder(cw, ch, n_slots=4)
    try:
        z = np.zeros(cw * ch, np.int16)
        grp_off, entries = V.sparsify(z, z[: cw * ch // 4], z[: cw * ch // 4], cw, ch)
        q, ia = np.full(nmb, 8, np.uint8), np.full(nmb, 255, np.uint8)
        bad = grp_off.copy()
        bad[3] = 5                              # not ascending / not closed by n_entries
        with pytest.raises(L.LeonError):
            dec.submit_sparse([L.make_sparse_picture(1, 0, bad, entries, 0, q, ia)], L.MEM_HOST)
        with pytest.raises(L.LeonError):        # more entries than a picture has coefficients
            dec.submit_sparse([L.make_sparse_picture(1, 0, grp_off, entries, cw * ch * 4, q, ia)], L.MEM_HOST)
        # offsets beyond the tile are masked inside the kernel: no fault, the picture still decodes
        ent = np.array([(0x3ff << 16) | 5, (0xffff << 16) | 7], np.uint32)
        go = np.zeros_like(grp_off)
        go[1:] = 2
        dec.submit_sparse([L.make_sparse_picture(1, 1, go, ent, 2, q, ia)], L.MEM_HOST)
        dec.sync()
        dec.read_planes(1)
    finally:
        dec.close()


@pytest.mark.parametrize("name", ["tiny_ip_32x32", "leon_synth_352x240", "ibbp_96x64", "slices5_ip_96x64"])
def test_stream_to_planes_through_the_native_front_end(L, O, name):
    """bytes -> leon_vlc_next_picture -> leon_submit_sparse, with the anchor bookkeeping of
    jsv.prototype.IDCT_GL (prev_pic_framebuffer, decoders/jsv.js:665) -- against the oracle fed
    with the densified tensors of the same parse."""
    import leon_vlc_ctypes as V
    with open(os.path.join(STREAMS, name + ".jsv"), "rb") as f:
        st = V.Stream(f.read(), threads=4)
    cw, ch = st.info.coded_width, st.info.coded_height
    qm = np.concatenate([np.frombuffer(bytes(st.info.intra_qm), np.uint8), np.frombuffer(bytes(st.info.non_intra_qm), np.uint8)])
    pics = []
    anchor_old, anchor_new = None, None
    n = 0
    while True:
        p = st.next_picture(dense=True)
        if p is None or n >= 14:
            break
        p["slot"] = n
        if p["type"] == 2:
            p["ref_fwd"], p["ref_bwd"] = anchor_new, None
        elif p["type"] == 3:
            p["ref_fwd"], p["ref_bwd"] = (anchor_old if anchor_old is not None else anchor_new), anchor_new
        else:
            p["ref_fwd"] = p["ref_bwd"] = None
        if p["type"] != 3:
            anchor_old, anchor_new = anchor_new, n
        pics.append(p)
        n += 1
    assert len(pics) >= 3
    _run(L, O, cw, ch, pics, qm=qm)
