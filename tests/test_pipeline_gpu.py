"""GPU: the native decode pipeline (include/leon_pipeline.h) -- stream bytes in, RGBA frames in device
memory out, K parser threads / one launch per picture type and dependency level across a window of GOPs /
fused display conversion -- against the oracle run picture by picture on the same stream."""
import os
import threading

import numpy as np
import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

STREAMS = os.path.join(ROOT, "tests", "golden", "streams")


@pytest.fixture(scope="module")
def L():
    import leon_ctypes
    leon_ctypes.load()
    return leon_ctypes


def oracle_frames(data):
    """{(gop, display_index): RGBA} by parsing with the native front end and decoding with the oracle"""
    import leon_vlc_ctypes as V
    from oracle import oracle_py as O
    st = V.Stream(data, threads=1)
    info = st.info
    cw, ch, fw, fh = info.coded_width, info.coded_height, info.frame_width, info.frame_height
    qm = np.concatenate([np.frombuffer(bytes(info.intra_qm), np.uint8), np.frombuffer(bytes(info.non_intra_qm), np.uint8)])
    out = {}
    gop = -1
    older = newer = None
    while True:
        p = st.next_picture(dense=True)
        if p is None:
            break
        if p["type"] == 1:
            gop += 1
            older = newer = None
        fwd = bwd = None
        if p["type"] == 2:
            fwd = newer
        elif p["type"] == 3:
            bwd, fwd = newer, (older if older is not None else newer)
        planes = O.decode_picture(p["type"], cw, ch, p["coef_y"], p["coef_cb"], p["coef_cr"], p["qscale"], p["intra"],
                                  repadd=p.get("repadd"), mb_dir=p.get("mb_dir"), mv_fwd=p.get("mv_fwd"), mv_bwd=p.get("mv_bwd"),
                                  qm=qm, ref_fwd=fwd, ref_bwd=bwd, coef_a=p.get("coef_a"))
        if p["type"] != 3:
            older, newer = newer, planes
        n3 = cw * ch * 3 // 2
        y, cb, cr = O.split_planes(planes[:n3], cw, ch)
        out[(gop, p["temporal_reference"])] = O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "cpu",
                                                               a=planes[n3:] if p.get("coef_a") is not None else None)
    return out


def differing_macroblocks(a, b):
    """[(macroblock row, column, differing pixels)] of two RGBA frames -- for assertion messages"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.ndim != 3:
        return "shapes %s / %s" % (a.shape, b.shape)
    d = np.any(a != b, axis=2)
    out = {}
    for y, x in zip(*np.nonzero(d)):
        out[(int(y) // 16, int(x) // 16)] = out.get((int(y) // 16, int(x) // 16), 0) + 1
    return sorted((r, c, n) for (r, c), n in out.items())


def run_pipeline(L, data, **kw):
    kw.setdefault("gpu_parser", False)          # the library's default is the GPU parser: tests/test_gpu_parser_gpu.py asks for it by name
    got = {}
    order = []

    def on_window(window, frames):
        for f in frames:
            got[(f["gop"], f["display_index"])] = L.read_frame(f)       # not `pipe`: unassigned while the constructor runs
            order.append((f["gop"], f["display_index"], f["ts_ms"]))
    pipe = L.Pipeline(data, on_window=on_window, **kw)
    try:
        pipe.wait()
        assert pipe.ended and pipe.error is None
        stats = pipe.stats()
    finally:
        pipe.close()
    return got, order, stats


def ibbp_stream(cw, ch, gops, seed, gop_qm=None, frame=None):
    """gop_qm: {GOP number: (intra matrix, non-intra matrix)} for GOPs whose sequence header carries matrices of its own;
    frame: (width, height) of the display crop when it is not the coded size"""
    import jsv_writer as W
    import synth as S
    rng = np.random.default_rng(seed)
    pics, starts = [], []
    for n in gops:
        starts.append(len(pics))
        for ptype, disp, f, b in S.gop_ibbp(n):
            t = S.make_picture(rng, cw, ch, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            t["display"] = disp
            pics.append(t)
    qm = {starts[g]: m for g, m in (gop_qm or {}).items()}
    fw, fh = frame or (cw, ch)
    return W.write_stream(pics, cw, ch, fw, fh, gop_starts=starts, gop_qm=qm)[0]


@pytest.mark.parametrize("name", ["leon_synth_352x240", "slices5_ip_96x64", "custom_intra_ip_48x32", "tiny_ip_32x32"])
def test_fixture_streams(L, name):
    data = open(os.path.join(STREAMS, name + ".jsv"), "rb").read()
    want = oracle_frames(data)
    got, order, stats = run_pipeline(L, data, parser_threads=3, gops_per_window=2)
    assert set(got) == set(want) and stats["pictures"] == len(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), "%s: frame %s differs" % (name, k)
    assert order == sorted(order)                       # GOP-major, display order inside a GOP


def test_yuva_stream(L):
    """a yuva stream (container flag `a`): the GOP shards are opened with the whole stream's flag
    (leon_vlc_open_shard) and the frames' A bytes are the decoded fourth component"""
    data = open(os.path.join(STREAMS, "yuva_ibbp_96x64.jsv"), "rb").read()
    want = oracle_frames(data)
    assert any((rgba.reshape(-1, 4)[:, 3] != 255).any() for rgba in want.values())
    got, order, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2)
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("window,threads,inflight", [(1, 1, 1), (3, 4, 2), (8, 2, 3)])
def test_ibbp_windows_and_loops(L, window, threads, inflight):
    """7 GOPs of different lengths (IBBP, closed): every window / thread / ring geometry gives the oracle's frames,
    and looping the stream continues the GOP numbering"""
    data = ibbp_stream(96, 64, [12, 6, 9, 12, 3, 12, 7], seed=4242)
    want = oracle_frames(data)
    got, order, stats = run_pipeline(L, data, parser_threads=threads, gops_per_window=window, windows_in_flight=inflight, loop=2)
    assert len(got) == 2 * len(want) and stats["gops"] == 14 and stats["pictures"] == 2 * len(want)
    for (g, d), img in want.items():
        assert np.array_equal(got[(g, d)], img), (g, d)
        assert np.array_equal(got[(g + 7, d)], img), (g + 7, d)
    ts = [t for _, _, t in order[:61]]
    assert ts == sorted(ts) and ts[1] - ts[0] == pytest.approx(40.0)      # 25 pictures/s


def test_last_macroblock_of_a_gop_shard(L):
    """seed 26: GOP 0 ends in a B picture whose last macroblock is two bytes of vectors; a shard cut at the key-map
    offset itself loses it (see leon_vlc_get_keymap)"""
    data = ibbp_stream(208, 112, [6, 9, 3, 12, 6, 9, 12], seed=26)
    want = oracle_frames(data)
    got, _, _ = run_pipeline(L, data, parser_threads=2, gops_per_window=3)
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def test_consumer_may_hold_windows(L):
    """frames stay valid until the window is released, from any thread; the pipeline waits for its ring"""
    data = ibbp_stream(96, 64, [6] * 6, seed=7)
    want = oracle_frames(data)
    held = []
    got = {}
    lock = threading.Lock()

    def on_window(window, frames):
        with lock:
            held.append((window, frames))
        return False                                     # keep it
    pipe = L.Pipeline(data, on_window=on_window, parser_threads=2, gops_per_window=2, windows_in_flight=2)
    try:
        import time
        released = 0
        while released < 3:
            time.sleep(0.05)
            with lock:
                batch, held[:] = list(held), []
            for window, frames in batch:
                for f in frames:
                    got[(f["gop"], f["display_index"])] = pipe.read_frame(f)
                pipe.release_window(window)
                released += 1
        pipe.wait()
    finally:
        pipe.close()
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_frame_widths_that_are_no_multiple_of_8(L, gpu_parser):
    """the reference crops to any width (player/easybits.player.js:2818); the fused display conversion needs
    frame_width % 8 == 0, so such streams take the unfused road inside the pipeline (planes for every picture + one
    conversion per picture): the fixture with a 90 x 60 crop, and a 61 x 45 crop of 64 x 48 (odd width: the conversion
    walks the frame with the reference's flat index, see k_rgba_twin)"""
    for data in (open(os.path.join(STREAMS, "ibbp_96x64.jsv"), "rb").read(),
                 ibbp_stream(64, 48, [6, 9, 3], seed=61, frame=(61, 45))):
        want = oracle_frames(data)
        got, order, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2, gpu_parser=gpu_parser)
        assert set(got) == set(want) and stats["pictures"] == len(want)
        for k in sorted(want):
            assert got[k].shape == want[k].shape
            bad = np.argwhere(got[k] != want[k])
            assert bad.size == 0, "frame %s differs in %d bytes, first at %s" % (k, len(bad), bad[0])


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_a_stream_that_is_still_arriving(L, gpu_parser):
    """leon_pipeline_create_partial + leon_pipeline_feed: the reference's decoder works on a growing buffer, stalls when
    it runs dry and goes on when a chunk is appended (features/bitreader.js:332-430, :135-189).  The pipeline is created
    on the container header + key map + one GOP; the other GOPs arrive in pieces that ignore GOP boundaries; frames
    come as their GOPs complete and equal the oracle's"""
    import time
    import leon_vlc_ctypes as V
    data = ibbp_stream(96, 64, [6, 9, 3, 12, 6], seed=77)
    want = oracle_frames(data)
    offs = V.Stream(data, threads=1).keymap()
    got, lock = {}, threading.Lock()

    def on_window(window, frames):
        with lock:
            for f in frames:
                got[(f["gop"], f["display_index"])] = L.read_frame(f)
    first = offs[1] + 3                     # GOP 0 complete (a shard takes the start code prefix of what follows along)
    buf = bytearray(len(data))
    buf[:first] = data[:first]
    pipe = L.Pipeline(bytes(buf), parser_threads=2, gops_per_window=1, gpu_parser=gpu_parser, on_window=on_window, valid_bytes=first)
    try:
        t0 = time.time()
        while time.time() - t0 < 20:
            with lock:
                if len([k for k in got if k[0] == 0]) == 6:
                    break
            time.sleep(0.01)
        with lock:
            assert sorted(k for k in got) == sorted(k for k in want if k[0] == 0), "GOP 0 decodes while the rest of the stream is still missing"
        time.sleep(0.05)
        with lock:
            assert all(k[0] == 0 for k in got)          # nothing of what has not arrived
        at = first
        for step in (500, 1, 1800, 700, 10 ** 9):
            n = min(step, len(data) - at)
            pipe.feed(at + n, data[at:at + n], at)
            at += n
            if at == len(data):
                break
        pipe.wait()
    finally:
        pipe.close()
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), (k, differing_macroblocks(got[k], want[k]))
    # feeding backwards or beyond the end is refused / ignored
    with pytest.raises(L.LeonError):
        L.Pipeline(data, valid_bytes=len(data) + 1)


def test_pipeline_errors(L):
    with pytest.raises(L.LeonError):
        L.Pipeline(b"\x00" * 64)
    # a damaged GOP: the run stops with an error instead of delivering garbage
    good = ibbp_stream(96, 64, [6, 6, 6], seed=9)
    bad = bytearray(good)
    import leon_vlc_ctypes as V
    offs = V.Stream(good, threads=1).keymap()
    for i in range(offs[1] + 60, offs[1] + 400):
        bad[i] = 0xFF
    pipe = L.Pipeline(bytes(bad), gops_per_window=1, parser_threads=1, max_gop_pictures=64)
    try:
        with pytest.raises(L.LeonError):
            pipe.wait()
    finally:
        pipe.close()


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
@pytest.mark.parametrize("which", ["intra", "non-intra"])
def test_a_later_sequence_header_with_other_matrices_is_refused(L, gpu_parser, which):
    """the pipeline dequantises with the matrices of the stream's first sequence header; every key-map GOP has a
    sequence header of its own (the reference reloads the matrices at each, decoders/jsv.js:540-558): a GOP whose
    header carries another matrix ends the run with an error instead of wrong pixels"""
    other = (np.arange(64, dtype=np.uint8) + 20)
    other[0] = 8
    qm = (other, None) if which == "intra" else (None, other)
    data = ibbp_stream(96, 64, [6, 6, 6], seed=11, gop_qm={1: qm})
    pipe = L.Pipeline(data, gops_per_window=1, parser_threads=1, max_gop_pictures=64, gpu_parser=gpu_parser)
    try:
        with pytest.raises(L.LeonError) as e:
            pipe.wait()
        assert "sequence header changes the %s quantiser matrix" % which in str(e.value)
    finally:
        pipe.close()
    # the same matrices in EVERY header are fine
    same = ibbp_stream(96, 64, [6, 6], seed=11, gop_qm={0: qm, 1: qm})
    want = oracle_frames(same)
    got, _, _ = run_pipeline(L, same, parser_threads=2, gops_per_window=2, gpu_parser=gpu_parser, max_gop_pictures=64)
    assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_gop_shards_partition_the_stream(L, gpu_parser):
    """two pipelines with shard_index 0 / 1 of 2 (one per GPU on a node; both on this box's device here)
    decode disjoint GOP sets whose union is the whole stream, frames keyed by the stream's GOP ids"""
    data = ibbp_stream(96, 64, [6, 9, 3, 12, 6], seed=31)
    want = oracle_frames(data)
    got = {}
    for r in range(2):
        part, _, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2, shard_index=r, shard_count=2, gpu_parser=gpu_parser)
        assert {g for g, _ in part} == {g for g in range(5) if g % 2 == r}
        assert not set(part) & set(got)
        got.update(part)
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    with pytest.raises(L.LeonError):
        L.Pipeline(data, shard_index=7, shard_count=8)       # more shards than GOPs: this one would be empty


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_destroy_while_running_and_with_windows_held(L, gpu_parser):
    """tearing a pipeline down mid-run -- parsers busy, windows in flight, windows delivered and never released --
    joins every thread and frees everything (no callback after close returns)"""
    data = ibbp_stream(96, 64, [6] * 8, seed=5)
    for hold in (False, True):
        for _ in range(3):
            seen = []
            pipe = L.Pipeline(data, on_window=lambda w, fr: (seen.append(w), False if hold else None)[1],
                              parser_threads=3, gops_per_window=2, windows_in_flight=2, loop=50, gpu_parser=gpu_parser)
            import time
            time.sleep(0.02)
            pipe.close()
            n = len(seen)
            time.sleep(0.05)
            assert len(seen) == n            # nothing arrives after close
    # and a normal run still works afterwards
    got, _, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=4, gpu_parser=gpu_parser)
    assert stats["gops"] == 8 and len(got) == 48
