"""GPU: size-independent properties at the BASELINE size (1920x1080, coded 1920x1088), where the
scalar oracle is too slow to check every picture: identities of the prediction, determinism,
independence of the batch composition, and equality of the two boundary formats."""
import numpy as np
import pytest

from helpers import planes_flat, hip_submit, hip_submit_sparse

pytestmark = pytest.mark.gpu
CW, CH = 1920, 1088
NMB = (CW // 16) * (CH // 16)


@pytest.fixture(scope="module")
def L():
    import leon_ctypes
    leon_ctypes.load()
    return leon_ctypes


@pytest.fixture(scope="module")
def S():
    import synth
    return synth


def _scene(S, seed):
    rng = np.random.default_rng(seed)
    y = S.smooth_scene(rng, CW, CH).astype(np.uint8)
    cb = S.smooth_scene(rng, CW // 2, CH // 2).astype(np.uint8)
    cr = S.smooth_scene(rng, CW // 2, CH // 2).astype(np.uint8)
    return y, cb, cr


def _empty(ptype, slot, mv=(0, 0), mvb=(0, 0), direction=3):
    z = lambda n: np.zeros(n, np.int16)
    t = {"type": ptype, "slot": slot, "coef_y": z(CW * CH), "coef_cb": z(CW * CH // 4), "coef_cr": z(CW * CH // 4),
         "qscale": np.full(NMB, 8, np.uint8), "intra": np.zeros(NMB, np.uint8), "repadd": np.zeros(NMB, np.uint8),
         "mv_fwd": np.tile(np.array(mv, np.int16), NMB), "ref_fwd": 0, "ref_bwd": None}
    if ptype == 3:
        t["mv_bwd"] = np.tile(np.array(mvb, np.int16), NMB)
        t["mb_dir"] = np.full(NMB, direction, np.uint8)
        t["ref_bwd"] = 1
    return t


def test_zero_residual_prediction_identities(L, S):
    """No coefficients: a P picture with a zero vector copies its reference; a full-pel vector
    translates it (interior); a half-pel vector averages neighbours with the standard rounding;
    a bidirectional picture of two references is their rounded-up mean."""
    a, b = _scene(S, 1), _scene(S, 2)
    dec = L.Decoder(CW, CH, n_slots=8)
    try:
        dec.write_planes(0, *a)
        dec.write_planes(1, *b)
        keep = []
        hip_submit(L, dec, _empty(2, 2), keep)
        hip_submit(L, dec, _empty(2, 3, mv=(2 * 6, 2 * -4)), keep)          # +6 px right, 4 px up (source offset)
        hip_submit(L, dec, _empty(2, 4, mv=(1, 1)), keep)                   # half-pel both ways
        hip_submit(L, dec, _empty(3, 5), keep)                              # bidirectional, zero vectors
        dec.sync()
        y, cb, cr = dec.read_planes(2)
        assert np.array_equal(y, a[0]) and np.array_equal(cb, a[1]) and np.array_equal(cr, a[2])
        y, cb, cr = dec.read_planes(3)
        assert np.array_equal(y[8:-8, 0:-16], a[0][4:-12, 6:-10])           # out[y][x] = ref[y-4][x+6]
        assert np.array_equal(cb[4:-4, 0:-8], a[1][2:-6, 3:-5])             # chroma: vector halved
        y, cb, cr = dec.read_planes(4)
        A = a[0].astype(np.int32)
        want = (A[:-1, :-1] + A[:-1, 1:] + A[1:, :-1] + A[1:, 1:] + 2) >> 2
        assert np.array_equal(y[:-1, :-1], want.astype(np.uint8))
        assert np.array_equal(cb, a[1])                                     # trunc(1/2) = 0: chroma full-pel
        y, cb, cr = dec.read_planes(5)
        assert np.array_equal(y, ((a[0].astype(np.int32) + b[0] + 1) >> 1).astype(np.uint8))
        assert np.array_equal(cr, ((a[2].astype(np.int32) + b[2] + 1) >> 1).astype(np.uint8))
    finally:
        dec.close()


def test_determinism_batch_independence_and_boundary_equality(L, S):
    """The same 1080p pictures: twice in a row, alone or next to other pictures in one launch, through
    dense planes or sparse group lists -- always the same planes."""
    import torch
    import leon_vlc_ctypes as V
    rng = np.random.default_rng(77)
    a, b = _scene(S, 3), _scene(S, 4)
    tens = [S.make_picture(rng, CW, CH, t) for t in (S.PIC_I, S.PIC_P, S.PIC_B, S.PIC_B)]
    dec = L.Decoder(CW, CH, n_slots=16)
    try:
        dec.write_planes(0, *a)
        dec.write_planes(1, *b)
        keep = []
        for i, t in enumerate(tens):
            t["slot"], t["ref_fwd"], t["ref_bwd"] = 2 + i, (0 if t["type"] != 1 else None), (1 if t["type"] == 3 else None)
            hip_submit(L, dec, t, keep)
        dec.sync()
        first = [planes_flat(*dec.read_planes(2 + i)) for i in range(4)]
        for i, t in enumerate(tens):                                         # again, into other slots, sparse
            t2 = dict(t, slot=6 + i)
            hip_submit_sparse(L, dec, t2, keep, CW, CH)
        dec.sync()
        for i in range(4):
            assert np.array_equal(planes_flat(*dec.read_planes(6 + i)), first[i]), "sparse boundary, picture %d" % i
        # one device-resident launch holding all four (mutually independent) pictures, sparse lists
        dev, pics = [], []
        for i, t in enumerate(tens):
            go, en = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], CW, CH)
            d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in t.items()
                 if isinstance(v, np.ndarray) and not k.startswith("coef")}
            d["grp_off"] = torch.from_numpy(go.view(np.int32)).cuda()
            d["entries"] = torch.from_numpy(en.view(np.int32)).cuda()
            dev.append(d)
            ptr = lambda k: d[k].data_ptr() if k in d else None
            pics.append(L.make_sparse_picture(t["type"], 10 + i, ptr("grp_off"), ptr("entries"), len(en), ptr("qscale"), ptr("intra"),
                                              ptr("repadd"), ptr("mv_fwd"), ptr("mv_bwd"), ptr("mb_dir"), ref_fwd_slot=0 if t["type"] != 1 else -1,
                                              ref_bwd_slot=1 if t["type"] == 3 else -1, device=True))
        torch.cuda.synchronize()
        batch = dec.batch_create_sparse(pics)
        for _ in range(2):
            dec.batch_run(batch)
        dec.sync()
        for i in range(4):
            assert np.array_equal(planes_flat(*dec.read_planes(10 + i)), first[i]), "batched sparse launch, picture %d" % i
        dec.batch_destroy(batch)
    finally:
        dec.close()
