"""GPU parity tests: libleon_hip.so (through the C ABI) against the CPU oracle, bit for bit.

The calls mirror how the reference drives its GL path: per picture the arrays that
jsv.prototype.IDCT_GL uploads (decoders/jsv.js:1204-1298) are handed to
leon_submit_picture / leon_submit_batch, and planes are read back where the
reference would bind the output textures.
"""
import numpy as np
import pytest

from helpers import oracle_decode_sequence, hip_submit, planes_flat

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


def _run_sequence(L, O, cw, ch, pics, n_slots=13, qm=None, refs=None):
    dec = L.Decoder(cw, ch, n_slots=n_slots)
    try:
        if qm is not None:
            dec.set_quant_matrices(qm[:64], qm[64:])
        refs = refs or {}
        for slot, flat in refs.items():
            y, cb, cr = O.split_planes(flat, cw, ch)
            dec.write_planes(slot, y, cb, cr)
        keep = []
        for t in pics:
            hip_submit(L, dec, t, keep)
        dec.sync()
        exp = oracle_decode_sequence(O, cw, ch, pics, refs, qm=qm)
        for t in pics:
            got = planes_flat(*dec.read_planes(t["slot"]))
            bad = np.nonzero(got != exp[t["slot"]])[0]
            assert bad.size == 0, "picture type %d slot %d: %d samples differ, first at %d (got %d want %d)" % (
                t["type"], t["slot"], bad.size, bad[0], got[bad[0]], exp[t["slot"]][bad[0]])
    finally:
        dec.close()


def _chain(S, rng, cw, ch, gop, **kw):
    """Tensors for a coded-order GOP; slot = display index."""
    pics = []
    for ptype, disp, f, b in gop:
        force = None
        if ptype == S.PIC_B and f is None:
            force = 2                      # closed-GOP leading B pictures: backward only
        t = S.make_picture(rng, cw, ch, ptype, force_dir=force, **kw)
        t["slot"] = disp
        t["ref_fwd"] = f if f is not None else (b if ptype == S.PIC_B else None)
        t["ref_bwd"] = b
        pics.append(t)
    return pics


def test_config2_i_only_352x240(L, O, S):
    rng = np.random.default_rng(0x4C454F4E)
    pics = _chain(S, rng, 352, 240, [(S.PIC_I, i, None, None) for i in range(4)])
    _run_sequence(L, O, 352, 240, pics)


def test_config3_1280x720_ippp_gop16(L, O, S):
    """BASELINE config 3: 1280x720, GOP = I + 15 P, per-macroblock vectors uniform in +-31 half-pel
    kept inside the picture, 10 % intra-in-P, 15 % skipped, random coded-block patterns."""
    rng = np.random.default_rng(720)
    _run_sequence(L, O, 1280, 720, _chain(S, rng, 1280, 720, S.gop_ippp(16)), n_slots=16)


def test_large_picture_4096x2304(L, O, S):
    """Well past the BASELINE size: 36 864 macroblocks, 1 152 luma groups per block row pair."""
    rng = np.random.default_rng(4096)
    gop = [(S.PIC_I, 0, None, None), (S.PIC_P, 2, 0, None), (S.PIC_B, 1, 0, 2)]
    _run_sequence(L, O, 4096, 2304, _chain(S, rng, 4096, 2304, gop), n_slots=4)


def test_ippp_in_picture_vectors(L, O, S):
    rng = np.random.default_rng(1)
    _run_sequence(L, O, 176, 144, _chain(S, rng, 176, 144, S.gop_ippp(6)))


def test_ippp_vectors_leave_picture(L, O, S):
    """Vectors that leave the coded picture: texel-granular CLAMP_TO_EDGE semantics."""
    rng = np.random.default_rng(2)
    _run_sequence(L, O, 64, 48, _chain(S, rng, 64, 48, S.gop_ippp(5), in_picture=False, mv_range=40))


def test_ibbp_gop(L, O, S):
    rng = np.random.default_rng(3)
    _run_sequence(L, O, 96, 64, _chain(S, rng, 96, 64, S.gop_ibbp(12)), n_slots=13)


def test_ibbp_vectors_leave_picture(L, O, S):
    rng = np.random.default_rng(4)
    _run_sequence(L, O, 48, 32, _chain(S, rng, 48, 32, S.gop_ibbp(9), in_picture=False, mv_range=63), n_slots=13)


def test_all_half_pel_phases_and_edges(L, O, S):
    """Every (H,V) phase at picture corners and centre, zero residual."""
    cw, ch = 64, 48
    mbw, mbh = 4, 3
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 256, size=cw * ch * 3 // 2).astype(np.uint8)
    pics = []
    slot = 1
    for mvh in (-5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5):
        for mvv in (-3, -2, -1, 0, 1, 2, 3):
            t = {"type": 2, "slot": slot, "ref_fwd": 0, "ref_bwd": None,
                 "coef_y": np.zeros((ch, cw), np.int16), "coef_cb": np.zeros((ch // 2, cw // 2), np.int16),
                 "coef_cr": np.zeros((ch // 2, cw // 2), np.int16),
                 "qscale": np.full(mbw * mbh, 8, np.uint8), "intra": np.zeros(mbw * mbh, np.uint8),
                 "repadd": np.zeros(mbw * mbh, np.uint8),
                 "mv_fwd": np.tile(np.array([mvh, mvv], np.int16), mbw * mbh)}
            pics.append(t)
            slot += 1
    _run_sequence(L, O, cw, ch, pics, n_slots=slot, refs={0: ref})


def test_custom_quant_matrices_and_zero_quirk(L, O, S):
    """Small custom matrix entries reach floor(.)==0 -> +1 (mpeg1video.js:22)."""
    rng = np.random.default_rng(6)
    qm = np.concatenate([rng.integers(1, 40, size=64), rng.integers(1, 40, size=64)]).astype(np.uint8)
    cw, ch = 64, 32
    pics = _chain(S, rng, cw, ch, S.gop_ippp(3))
    for t in pics:
        for k in ("coef_y", "coef_cb", "coef_cr"):
            c = rng.integers(-3, 4, size=t[k].shape).astype(np.int16)
            c[rng.random(c.shape) > 0.4] = 0
            t[k] = c
        t["qscale"] = rng.integers(1, 4, size=t["qscale"].shape).astype(np.uint8)
    _run_sequence(L, O, cw, ch, pics, qm=qm)


@pytest.mark.parametrize("amp", [255, 2047, 32767])
def test_garbage_levels_full_int16_range(L, O, S, amp):
    """Outside the domain of any real stream: the int16 hand-off wraps/saturates
    exactly like _B()/_E() + the RGBA8 store (mpeg1video.js:18)."""
    rng = np.random.default_rng(7 + amp)
    cw, ch = 48, 32
    pics = _chain(S, rng, cw, ch, S.gop_ibbp(6))
    for t in pics:
        for k in ("coef_y", "coef_cb", "coef_cr"):
            c = rng.integers(-amp, amp + 1, size=t[k].shape).astype(np.int16)
            c[rng.random(c.shape) > 0.5] = 0
            t[k] = c
        t["qscale"] = rng.integers(0, 32, size=t["qscale"].shape).astype(np.uint8)
    _run_sequence(L, O, cw, ch, pics)


def test_ragged_width_partial_block_groups(L, O, S):
    """Widths that are not a multiple of 64 (luma) / 128 (chroma groups): 16, 80, 208."""
    for cw, ch in ((16, 16), (80, 32), (208, 48)):
        rng = np.random.default_rng(cw)
        _run_sequence(L, O, cw, ch, _chain(S, rng, cw, ch, S.gop_ibbp(6)))


def test_rgba_both_flavours(L, O, S):
    rng = np.random.default_rng(8)
    for (cw, ch, fw, fh) in ((64, 48, 64, 48), (64, 48, 61, 45), (352, 240, 352, 240), (32, 32, 30, 31)):
        dec = L.Decoder(cw, ch, fw, fh, n_slots=2)
        try:
            y = rng.integers(0, 256, size=(ch, cw)).astype(np.uint8)
            cb = rng.integers(0, 256, size=(ch // 2, cw // 2)).astype(np.uint8)
            cr = rng.integers(0, 256, size=(ch // 2, cw // 2)).astype(np.uint8)
            dec.write_planes(1, y, cb, cr)
            got = dec.convert_rgba(1, L.RGB_CPU_TWIN)
            exp = O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "cpu")
            assert np.array_equal(got, exp), "cpu twin %s: %d bytes differ" % ((cw, ch, fw, fh), (got != exp).sum())
            got = dec.convert_rgba(1, L.RGB_GL)
            exp = O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "gl")
            assert np.array_equal(got, exp), "gl %s: %d bytes differ" % ((cw, ch, fw, fh), (got != exp).sum())
        finally:
            dec.close()


def test_rgba_golden_from_reference(L):
    """The fixture was produced by the reference's own YCbCrToRGBA under Node."""
    from helpers import load_golden, b64
    g = load_golden("rgb_ycbcrtorgba.json")
    for s in g["sets"]:
        cw, ch, fw, fh = s["coded_w"], s["coded_h"], s["frame_w"], s["frame_h"]
        dec = L.Decoder(cw, ch, fw, fh, n_slots=1)
        try:
            dec.write_planes(0, b64(s["y"]).reshape(ch, cw), b64(s["cb"]).reshape(ch // 2, cw // 2),
                             b64(s["cr"]).reshape(ch // 2, cw // 2))
            got = dec.convert_rgba(0, L.RGB_CPU_TWIN)
            exp = b64(s["rgba"]).reshape(fh, fw, 4)
            assert np.array_equal(got, exp), s["name"]
        finally:
            dec.close()


def test_mc_golden_from_reference(L):
    """copyMacroblock outputs (decoders/jsv.js:895-1129) reproduced by the HIP predictor."""
    from helpers import load_golden, b64
    g = load_golden("mc_copymacroblock.json")
    cw, ch = g["coded_w"], g["coded_h"]
    mbw, mbh = cw // 16, ch // 16
    cases = g["cases"]
    dec = L.Decoder(cw, ch, n_slots=len(cases) + 1)
    try:
        dec.write_planes(0, b64(g["ref_y"]).reshape(ch, cw), b64(g["ref_cb"]).reshape(ch // 2, cw // 2),
                         b64(g["ref_cr"]).reshape(ch // 2, cw // 2))
        keep = []
        zero_y = np.zeros((ch, cw), np.int16)
        zero_c = np.zeros((ch // 2, cw // 2), np.int16)
        for i, c in enumerate(cases):
            mv = np.zeros(mbw * mbh * 2, np.int16)
            mb = c["mbRow"] * mbw + c["mbCol"]
            mv[2 * mb], mv[2 * mb + 1] = c["mvH"], c["mvV"]
            p = L.make_picture(L.PIC_P, i + 1, zero_y, zero_c, zero_c, np.full(mbw * mbh, 8, np.uint8),
                               np.zeros(mbw * mbh, np.uint8), repadd=np.zeros(mbw * mbh, np.uint8), mv_fwd=mv,
                               ref_fwd_slot=0, keep=keep)
            dec.submit_picture(p)
        dec.sync()
        for i, c in enumerate(cases):
            y, cb, cr = dec.read_planes(i + 1)
            r, q = c["mbRow"], c["mbCol"]
            assert np.array_equal(y[16 * r:16 * r + 16, 16 * q:16 * q + 16].ravel(), b64(c["y"])), (c["mvH"], c["mvV"])
            assert np.array_equal(cb[8 * r:8 * r + 8, 8 * q:8 * q + 8].ravel(), b64(c["cb"])), (c["mvH"], c["mvV"])
            assert np.array_equal(cr[8 * r:8 * r + 8, 8 * q:8 * q + 8].ravel(), b64(c["cr"])), (c["mvH"], c["mvV"])
    finally:
        dec.close()


def test_slot_ring_semantics(L):
    dec = L.Decoder(32, 32, n_slots=13)
    try:
        got = [dec.acquire_slot() for _ in range(13)]
        assert got == list(range(13))
        with pytest.raises(L.LeonError) as e:
            dec.acquire_slot()                     # = throw "no free render buffers" (jsv.js:1175)
        assert e.value.code == L.ERR_NO_FREE_SLOT and "no free render buffers" in str(e.value)
        dec.release_slot(5)
        assert dec.acquire_slot() == 5
        dec.free_decoded_slots()                   # = GLfreeDecodedBuffers (jsv.js:1160)
        assert dec.acquire_slot() == 0
    finally:
        dec.close()


def test_argument_errors(L):
    with pytest.raises(L.LeonError):
        L.Decoder(100, 64)                         # not a multiple of 16
    dec = L.Decoder(32, 32, n_slots=2)
    try:
        z = np.zeros((32, 32), np.int16)
        zc = np.zeros((16, 16), np.int16)
        m = np.zeros(4, np.uint8)
        with pytest.raises(L.LeonError):           # P picture without a reference slot
            dec.submit_picture(L.make_picture(L.PIC_P, 0, z, zc, zc, m, m, repadd=m, mv_fwd=np.zeros(8, np.int16)))
        with pytest.raises(L.LeonError):           # out slot out of range
            dec.submit_picture(L.make_picture(L.PIC_I, 7, z, zc, zc, m, m))
    finally:
        dec.close()


def test_batch_independence_is_checked_for_any_size(L):
    """A picture that writes or reads a slot another picture of the same batch writes is refused --
    for batches beyond 64 pictures and for prepared batches too (leon_batch_create), in O(n)."""
    n = 200
    dec = L.Decoder(32, 32, n_slots=n + 2)
    try:
        z = np.zeros((32, 32), np.int16)
        zc = np.zeros((16, 16), np.int16)
        m = np.zeros(4, np.uint8)
        mv = np.zeros(8, np.int16)
        keep = []
        mk_i = lambda slot: L.make_picture(L.PIC_I, slot, z, zc, zc, m, m, keep=keep)
        mk_p = lambda slot, ref: L.make_picture(L.PIC_P, slot, z, zc, zc, m, m, repadd=m, mv_fwd=mv, ref_fwd_slot=ref, keep=keep)
        mk_b = lambda slot, f, b: L.make_picture(L.PIC_B, slot, z, zc, zc, m, m, repadd=m, mv_fwd=mv, mv_bwd=mv, mb_dir=m,
                                                 ref_fwd_slot=f, ref_bwd_slot=b, keep=keep)
        good = [mk_i(k) for k in range(n)]
        dec.submit_batch(good, L.MEM_HOST)                       # independent: accepted
        for bad_tail, what in (([mk_i(7)], "both write slot 7"),                    # duplicate output, far apart
                               ([mk_p(n, 150)], "reads slot 150"),                  # P reads what picture 150 writes
                               ([mk_b(n, n + 1, 199)], "reads slot 199")):          # B backward reference written in-batch
            for call in (lambda pics: dec.submit_batch(pics, L.MEM_HOST), dec.batch_create):
                with pytest.raises(L.LeonError) as e:
                    call(good + bad_tail)
                assert e.value.code == L.ERR_INVALID and "depend on each other" in str(e.value) and what in str(e.value), str(e.value)
        # the anchor before the pictures that read it is the classic mistake: refused as well
        with pytest.raises(L.LeonError):
            dec.batch_create([mk_i(0), mk_p(1, 0)])
        dec.sync()
    finally:
        dec.close()


def test_coded_size_beyond_the_format_is_refused(L):
    with pytest.raises(L.LeonError) as e:
        L.Decoder(4112, 64)
    assert "4096" in str(e.value)


def test_device_batch_equals_single_submits_1080p(L, O, S):
    """Full-size property test: a device-resident batch of independent 1080p pictures
    gives the same planes as one-by-one host submits, and spot rows match the oracle."""
    import torch
    cw, ch = 1920, 1088
    rng = np.random.default_rng(9)
    ref_a = S.smooth_scene(rng, cw, ch).astype(np.uint8)
    refs = [planes_flat(np.roll(ref_a, 7 * k, axis=1), np.roll(ref_a[::2, ::2], k, axis=0), ref_a[1::2, 1::2])
            for k in range(2)]
    tens = [S.make_picture(rng, cw, ch, S.PIC_B), S.make_picture(rng, cw, ch, S.PIC_P),
            S.make_picture(rng, cw, ch, S.PIC_I), S.make_picture(rng, cw, ch, S.PIC_B)]
    dec = L.Decoder(cw, ch, n_slots=12)
    try:
        for s, flat in enumerate(refs):
            dec.write_planes(s, *O.split_planes(flat, cw, ch))
        keep, dev, pics = [], [], []
        for i, t in enumerate(tens):
            t["slot"], t["ref_fwd"], t["ref_bwd"] = 2 + i, 0, 1
            hip_submit(L, dec, t, keep)
        dec.sync()
        single = [planes_flat(*dec.read_planes(2 + i)) for i in range(len(tens))]
        for i, t in enumerate(tens):
            d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in t.items() if isinstance(v, np.ndarray)}
            dev.append(d)
            ptr = lambda k: d[k].data_ptr() if k in d else None
            pics.append(L.make_picture(t["type"], 6 + i, ptr("coef_y"), ptr("coef_cb"), ptr("coef_cr"), ptr("qscale"),
                                       ptr("intra"), ptr("repadd"), ptr("mv_fwd"), ptr("mv_bwd"), ptr("mb_dir"),
                                       ref_fwd_slot=0, ref_bwd_slot=1, device=True))
        torch.cuda.synchronize()
        dec.submit_batch(pics, L.MEM_DEVICE)
        dec.sync()
        for i in range(len(tens)):
            assert np.array_equal(planes_flat(*dec.read_planes(6 + i)), single[i]), "batch picture %d" % i
        # oracle on the P picture only (a few seconds on one core)
        t = tens[1]
        exp = O.decode_picture(t["type"], cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                               repadd=t["repadd"], mv_fwd=t["mv_fwd"], ref_fwd=refs[0])
        assert np.array_equal(single[1], exp)
    finally:
        dec.close()


def test_overlapped_conversion_is_ordered_against_slot_reuse(L, O, S):
    """leon_set_overlap_convert: the RGBA conversion runs on a second stream; a later submit that
    overwrites a slot still being converted must wait for it."""
    import torch
    cw, ch = 352, 240
    rng = np.random.default_rng(11)
    first = [S.make_picture(rng, cw, ch, S.PIC_I) for _ in range(6)]
    second = [S.make_picture(rng, cw, ch, S.PIC_I) for _ in range(6)]
    dec = L.Decoder(cw, ch, n_slots=6)
    try:
        dec.set_overlap_convert(True)
        keep = []
        for i, t in enumerate(first):
            t["slot"], t["ref_fwd"], t["ref_bwd"] = i, None, None
            hip_submit(L, dec, t, keep)
        out = torch.zeros((6, ch, cw, 4), dtype=torch.uint8, device="cuda")
        for rep in range(3):                    # several rounds to give a race a chance to show
            dec.convert_rgba_batch(np.arange(6, dtype=np.int32), out.data_ptr())
            for i, t in enumerate(second if rep % 2 == 0 else first):
                t["slot"], t["ref_fwd"], t["ref_bwd"] = i, None, None
                hip_submit(L, dec, t, keep)
            dec.sync()
            src = first if rep % 2 == 0 else second
            got = out.cpu().numpy()
            for i, t in enumerate(src):
                planes = O.split_planes(O.decode_picture(1, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"]), cw, ch)
                assert np.array_equal(got[i], O.ycbcr_to_rgba(*planes, cw, cw, ch, "cpu")), (rep, i)
    finally:
        dec.close()


def test_descriptor_ring_wraps_without_draining_the_stream(L, O, S):
    """Ad-hoc batches take their picture descriptors from a ring of 65536 (leon_hip.cpp reserve_descs); a range is reused
    once the launches that read it a lap ago have finished (an event per 2048 descriptors) -- round 3 waited for the whole
    stream at the wrap, which cost the pipeline five windows' time every 65536 pictures.  72 batches of 1000 pictures
    queued without a sync in between (the ring wraps once), two contents alternating by batch: every slot must end up
    with the content of the LAST batch that wrote it."""
    import torch
    cw = ch = 32
    n = 1000
    rng = np.random.default_rng(31)
    tens = [S.make_picture(rng, cw, ch, S.PIC_I) for _ in range(2)]
    dec = L.Decoder(cw, ch, n_slots=n)
    try:
        dev = [{k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in t.items() if isinstance(v, np.ndarray)} for t in tens]
        torch.cuda.synchronize()

        def batch(which):
            d = dev[which]
            return [L.make_picture(S.PIC_I, i, d["coef_y"].data_ptr(), d["coef_cb"].data_ptr(), d["coef_cr"].data_ptr(),
                                   d["qscale"].data_ptr(), d["intra"].data_ptr(), device=True) for i in range(n)]
        batches = [batch(0), batch(1)]
        laps = 72
        for k in range(laps):
            dec.submit_batch(batches[k & 1], L.MEM_DEVICE)
        dec.sync()
        last = (laps - 1) & 1
        t = tens[last]
        exp = O.decode_picture(t["type"], cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"])
        for slot in (0, 1, 511, 999):
            assert np.array_equal(planes_flat(*dec.read_planes(slot)), exp), slot
        # and the ring goes on after the wrap
        dec.submit_batch(batches[1 - last], L.MEM_DEVICE)
        dec.sync()
        t = tens[1 - last]
        exp = O.decode_picture(t["type"], cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"])
        assert np.array_equal(planes_flat(*dec.read_planes(500)), exp)
    finally:
        dec.close()


@pytest.mark.parametrize("fused", [False, True], ids=["k_recon", "k_recon_display"])
def test_pictures_of_one_batch_with_different_matrix_sets(L, O, S, fused):
    """leon_add_quant_matrices + leon_picture.qm_set (ABI 3): the reference reloads both matrices at every sequence header
    (decoders/jsv.js:540-558); a batch may hold pictures of different sequences -- each is dequantised with ITS set, in one
    launch.  Three I pictures and three P pictures, sets 0 (leon_set_quant_matrices), 1 and 2, against the oracle with the
    matrices of each; the same matrices registered twice give the same id; an unknown set is refused."""
    import torch
    cw, ch = 96, 64
    rng = np.random.default_rng(21)
    mats = [np.concatenate([rng.integers(1, 60, size=64), rng.integers(1, 60, size=64)]).astype(np.uint8) for _ in range(3)]
    for m in mats:
        m[0] = 8
    dec = L.Decoder(cw, ch, n_slots=12)
    try:
        dec.set_quant_matrices(mats[0][:64], mats[0][64:])
        ids = [0, dec.add_quant_matrices(mats[1][:64], mats[1][64:]), dec.add_quant_matrices(mats[2][:64], mats[2][64:])]
        assert ids == [0, 1, 2] and dec.add_quant_matrices(mats[1][:64], mats[1][64:]) == 1
        tens_i = [S.make_picture(rng, cw, ch, S.PIC_I) for _ in range(3)]
        tens_p = [S.make_picture(rng, cw, ch, S.PIC_P) for _ in range(3)]
        frames = torch.zeros((6, ch, cw, 4), dtype=torch.uint8, device="cuda") if fused else None
        keep = []

        def batch(tens, first_slot, refs, f0):
            pics = []
            for i, t in enumerate(tens):
                d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in t.items() if isinstance(v, np.ndarray)}
                keep.append(d)
                ptr = lambda k, d=d: d[k].data_ptr() if k in d else None
                pics.append(L.make_picture(t["type"], first_slot + i, ptr("coef_y"), ptr("coef_cb"), ptr("coef_cr"), ptr("qscale"), ptr("intra"),
                                           ptr("repadd"), ptr("mv_fwd"), None, None, ref_fwd_slot=-1 if refs is None else refs + i, device=True,
                                           qm_set=ids[i], rgba_out=frames[f0 + i].data_ptr() if fused else None))
            torch.cuda.synchronize()
            dec.submit_batch(pics, L.MEM_DEVICE)
        batch(tens_i, 0, None, 0)
        batch(tens_p, 3, 0, 3)
        dec.sync()
        for i in range(3):
            exp_i = O.decode_picture(S.PIC_I, cw, ch, tens_i[i]["coef_y"], tens_i[i]["coef_cb"], tens_i[i]["coef_cr"], tens_i[i]["qscale"], tens_i[i]["intra"], qm=mats[i])
            assert np.array_equal(planes_flat(*dec.read_planes(i)), exp_i), "I picture with matrix set %d" % ids[i]
            t = tens_p[i]
            exp_p = O.decode_picture(S.PIC_P, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], repadd=t["repadd"],
                                     mv_fwd=t["mv_fwd"], qm=mats[i], ref_fwd=exp_i)
            assert np.array_equal(planes_flat(*dec.read_planes(3 + i)), exp_p), "P picture with matrix set %d" % ids[i]
            if fused:
                y, cb, cr = O.split_planes(exp_p, cw, ch)
                assert np.array_equal(frames[3 + i].cpu().numpy(), O.ycbcr_to_rgba(y, cb, cr, cw, cw, ch, "cpu"))
        # the sets really differ in their effect
        other = O.decode_picture(S.PIC_I, cw, ch, tens_i[1]["coef_y"], tens_i[1]["coef_cb"], tens_i[1]["coef_cr"], tens_i[1]["qscale"], tens_i[1]["intra"], qm=mats[0])
        assert not np.array_equal(other, planes_flat(*dec.read_planes(1)))
        bad = L.make_picture(S.PIC_I, 9, keep[0]["coef_y"].data_ptr(), keep[0]["coef_cb"].data_ptr(), keep[0]["coef_cr"].data_ptr(),
                             keep[0]["qscale"].data_ptr(), keep[0]["intra"].data_ptr(), device=True, qm_set=7)
        with pytest.raises(L.LeonError):
            dec.submit_batch([bad], L.MEM_DEVICE)
    finally:
        dec.close()
