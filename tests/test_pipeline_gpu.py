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


def oracle_frames(data, detail=None):
    """{(gop, display_index): RGBA} by parsing with the native front end and decoding with the oracle.  The quantiser
    matrices are those of the sequence header in force (the reference reloads them at each, decoders/jsv.js:540-558).
    detail (a dict): filled with {(gop, display_index): the picture's boundary tensors, its planes and the planes it
    predicted from} -- what explain_mismatch() holds a capture of the pipeline against."""
    import leon_vlc_ctypes as V
    from oracle import oracle_py as O
    st = V.Stream(data, threads=1)
    info = st.info
    cw, ch, fw, fh = info.coded_width, info.coded_height, info.frame_width, info.frame_height
    matrices = lambda i: np.concatenate([np.frombuffer(bytes(i.intra_qm), np.uint8), np.frombuffer(bytes(i.non_intra_qm), np.uint8)])
    qm = matrices(info)
    out = {}
    gop = -1
    older = newer = None
    while True:
        p = st.next_picture(dense=True)
        if p is None:
            break
        if p["new_sequence"]:
            qm = matrices(st.refresh_info())
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
        if detail is not None:
            detail[(gop, p["temporal_reference"])] = {"pic": p, "planes": planes, "fwd": fwd, "bwd": bwd, "qm": qm.copy()}
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


# ---- a wrong frame comes with the inputs that produced it ----------------------------------------------------------------------
# Round 3 saw B pictures of small pipelines come out wrong in whole macroblocks, late in a process that had freed physically
# contiguous slot rings -- and had nothing but the frame to look at.  The pipelines of this file run
# with LEON_DEBUG_CAPTURE (csrc/leon_pipeline_impl.h): after every dependency level the planes each picture wrote and predicted
# from, after the last one every GOP's device arena (maps, group offsets, entry lists as the kernels left them).  A frame that
# differs from the oracle's is then explained: which of its inputs held other bytes than the oracle's -- maps, lists, a
# reference slot -- and the capture is kept under gpurun_out/flake/ (which travels back from the GPU box).  No mismatch:
# the capture is deleted.

CAPTURE_ROOT = os.path.join(ROOT, "gpurun_out", "capture")
_capture_no = [0]


def start_capture():
    import shutil
    _capture_no[0] += 1
    d = os.path.join(CAPTURE_ROOT, "%d_%d" % (os.getpid(), _capture_no[0]))
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    os.environ["LEON_DEBUG_CAPTURE"] = d          # read by leon_pipeline_create*
    return d


def end_capture(d, keep_as=None):
    import shutil
    os.environ.pop("LEON_DEBUG_CAPTURE", None)
    if keep_as and os.path.isdir(d):
        dst = os.path.join(ROOT, "gpurun_out", "flake", keep_as)
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.move(d, dst)
        return dst
    shutil.rmtree(d, ignore_errors=True)
    return None


def read_capture_index(d):
    """[{'window':, 'lane':, 'gop':, 'tref':, ... offsets ...}], geometry"""
    pics, geom = [], {}
    for w in sorted(os.listdir(d)):
        idx = os.path.join(d, w, "index.txt")
        if not os.path.exists(idx):
            continue
        for line in open(idx):
            kind, *kv = line.split()
            rec = dict(x.split("=", 1) for x in kv)
            if kind == "geom":
                geom = rec
            elif kind == "pic":
                rec = {k: int(v) for k, v in rec.items()}
                rec["window"] = w
                pics.append(rec)
    return pics, geom


def explain_mismatch(d, key, detail, got=None, want=None):
    """text: which inputs of frame `key` = (gop, display index) differ, in the capture `d`, from what the oracle decoded from"""
    lines = []
    try:
        pics, geom = read_capture_index(d)
        hit = [r for r in pics if (r["gop"], r["tref"]) == tuple(key)]
        if not hit:
            return "no capture of frame %s in %s" % (key, d)
        r, ref = hit[-1], detail[tuple(key)]
        p = ref["pic"]
        mbs, ng = int(geom["mbs"]), int(geom["n_groups"])
        arena = np.fromfile(os.path.join(d, r["window"], "arena_%d.bin" % r["lane"]), np.uint8)
        lines.append("frame %s: type %d, level %d, window %s lane %d, slots out/fwd/bwd %d/%d/%d, %s parser" %
                     (key, r["type"], r["level"], r["window"], r["lane"], r["out"], r["fwd"], r["bwd"], "GPU" if geom.get("gpu_parser") == "1" else "host"))
        if got is not None and want is not None:
            lines.append("  differing macroblocks (row, col, pixels): %s" % (differing_macroblocks(got, want),))
        coded = None
        for name, dt, n in (("qscale", np.uint8, mbs), ("intra", np.uint8, mbs), ("repadd", np.uint8, mbs), ("mb_dir", np.uint8, mbs),
                            ("mv_fwd", np.int16, 2 * mbs), ("mv_bwd", np.int16, 2 * mbs)):
            if r[name] < 0 or p.get(name) is None:
                continue
            have = arena[r[name]:r[name] + n * np.dtype(dt).itemsize].view(dt)
            exp = np.asarray(p[name], dt).reshape(-1)
            bad = np.nonzero(have != exp)[0]
            if dt == np.int16:
                bad = np.unique(bad // 2)
            lines.append("  map %-7s: %d macroblocks differ from the host parser's%s" % (name, len(bad), (" -- " + str(bad[:16].tolist())) if len(bad) else ""))
        go = arena[r["grp_off"]:r["grp_off"] + 4 * (ng + 1)].view(np.uint32)
        ego = np.asarray(p["grp_off"], np.uint32)
        ent = arena[r["entries"]:r["entries"] + 4 * int(go[-1])].view(np.uint32) if go[-1] < (1 << 24) else np.zeros(0, np.uint32)
        eent = np.asarray(p["entries"], np.uint32)
        bad_g = [g for g in range(ng) if sorted(ent[go[g]:go[g + 1]].tolist()) != sorted(eent[ego[g]:ego[g + 1]].tolist())]
        lines.append("  entry lists: %d of %d groups differ from the host parser's (as sets)%s" % (len(bad_g), ng, (" -- " + str(bad_g[:16])) if bad_g else ""))
        for which in ("fwd", "bwd", "out"):
            f = os.path.join(d, r["window"], "L%d_g%d_t%d_%s.planes" % (r["level"], r["lane"], r["tref"], which))
            exp = ref["planes"] if which == "out" else ref[which]
            if not os.path.exists(f) or exp is None:
                continue
            have = np.fromfile(f, np.uint8)
            n = min(len(have), len(exp))
            bad = np.nonzero(have[:n] != np.asarray(exp, np.uint8)[:n])[0]
            lines.append("  %s slot planes after the level: %d of %d bytes differ from the oracle's%s" %
                         (which, len(bad), n, (" (first at %d, last at %d)" % (bad[0], bad[-1])) if len(bad) else ""))
    except Exception as e:          # the explanation must never hide the failure it explains
        lines.append("  (explain_mismatch itself failed: %r)" % (e,))
    return "\n".join(lines)


def assert_frames(got, want, detail, capture, name):
    """got == want frame by frame; a difference is explained from the capture, which is then kept"""
    bad = [k for k in sorted(want) if k not in got or got[k].shape != want[k].shape or not np.array_equal(got[k], want[k])]
    if not bad:
        end_capture(capture)
        return
    text = "\n".join(explain_mismatch(capture, k, detail, got.get(k), want[k]) for k in bad[:3])
    kept = end_capture(capture, keep_as=name)
    raise AssertionError("%d frames differ from the oracle: %s\n%s\ncapture kept in %s" % (len(bad), bad[:8], text, kept))


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
        detail = {}
        want = oracle_frames(data, detail)
        cap = start_capture()
        try:
            got, order, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2, gpu_parser=gpu_parser)
        except BaseException:
            end_capture(cap)
            raise
        assert set(got) == set(want) and stats["pictures"] == len(want)
        assert_frames(got, want, detail, cap, "frame_widths_%s" % ("gpu" if gpu_parser else "host"))


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_a_stream_that_is_still_arriving(L, gpu_parser):
    """leon_pipeline_create_partial + leon_pipeline_feed: the reference's decoder works on a growing buffer, stalls when
    it runs dry and goes on when a chunk is appended (features/bitreader.js:332-430, :135-189).  The pipeline is created
    on the container header + key map + one GOP; the other GOPs arrive in pieces that ignore GOP boundaries; frames
    come as their GOPs complete and equal the oracle's"""
    import time
    import leon_vlc_ctypes as V
    data = ibbp_stream(96, 64, [6, 9, 3, 12, 6], seed=77)
    detail = {}
    want = oracle_frames(data, detail)
    offs = V.Stream(data, threads=1).keymap()
    got, lock = {}, threading.Lock()
    cap = start_capture()

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
    except BaseException:
        end_capture(cap)
        raise
    finally:
        pipe.close()
    assert set(got) == set(want)
    assert_frames(got, want, detail, cap, "still_arriving_%s" % ("gpu" if gpu_parser else "host"))
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
@pytest.mark.parametrize("which", ["intra", "non-intra", "both"])
def test_a_later_sequence_header_with_other_matrices_decodes_to_the_oracles_frames(L, gpu_parser, which):
    """every key-map GOP has a sequence header of its own and the reference reloads both quantiser matrices at each
    (decoders/jsv.js:540-558): a stream whose GOPs carry DIFFERENT matrices decodes -- every picture with the matrices of
    its own sequence (leon_add_quant_matrices / leon_picture.qm_set), also when GOPs of different sequences share a window
    and a launch.  (Round 3 refused such a stream.)"""
    other = (np.arange(64, dtype=np.uint8) + 20)
    other[0] = 8
    third = np.full(64, 40, np.uint8)
    third[0] = 8
    qm1 = {"intra": (other, None), "non-intra": (None, other), "both": (other, third)}[which]
    qm3 = {"intra": (third, None), "non-intra": (None, third), "both": (third, other)}[which]
    data = ibbp_stream(96, 64, [6, 6, 9, 6, 3], seed=11, gop_qm={1: qm1, 3: qm3, 4: qm1})
    detail = {}
    want = oracle_frames(data, detail)
    assert len({d["qm"].tobytes() for d in detail.values()}) == 3          # the oracle really switched matrices
    plain = oracle_frames(ibbp_stream(96, 64, [6, 6, 9, 6, 3], seed=11))
    assert any(not np.array_equal(want[k], plain[k]) for k in want if k[0] == 1)      # and they matter to the pixels
    for window in (1, 4):
        cap = start_capture()
        try:
            got, _, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=window, gpu_parser=gpu_parser, max_gop_pictures=64)
        except BaseException:
            end_capture(cap)
            raise
        assert set(got) == set(want) and stats["pictures"] == len(want)
        assert_frames(got, want, detail, cap, "matrices_%s_%d" % (which, window))


@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_a_sequence_header_that_changes_the_picture_size_is_refused(L, gpu_parser):
    """the one thing of a later sequence header the pipeline cannot honour: its rings are made for one size"""
    import jsv_writer as W
    import synth as S

    def one_gop(cw, ch, seed):
        rng = np.random.default_rng(seed)
        pics = []
        for ptype, disp, f, b in S.gop_ibbp(6):
            t = S.make_picture(rng, cw, ch, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            t["display"] = disp
            pics.append(t)
        return W.write_stream(pics, cw, ch, cw, ch, gop_starts=[0])[0]
    data, _ = W.merge_gops([one_gop(96, 64, 1), one_gop(64, 48, 2), one_gop(96, 64, 3)], 96, 64)
    pipe = L.Pipeline(data, gops_per_window=1, parser_threads=1, max_gop_pictures=64, gpu_parser=gpu_parser)
    try:
        with pytest.raises(L.LeonError) as e:
            pipe.wait()
        assert "changes the picture size" in str(e.value)
    finally:
        pipe.close()


def test_default_parser_falls_back_to_the_host_beyond_the_gpu_parsers_limits(L, monkeypatch):
    """a caller who never asked for the GPU parser must not be refused for its limits (group counters beyond
    LDS, shards of 2^28 bytes): LEON_PIPELINE_PARSER_DEFAULT then decodes on the parser threads, an explicit
    LEON_PIPELINE_PARSER_GPU keeps its error.  The limit is mocked (LEON_DEBUG_GPU_PARSER_LIMIT = groups per picture)."""
    data = ibbp_stream(96, 64, [6, 6], seed=5)
    want = oracle_frames(data)
    p = L.Pipeline(data, gops_per_window=2, parser_threads=1)
    try:
        assert p.info.gpu_parser == 1
        p.wait()
    finally:
        p.close()
    monkeypatch.setenv("LEON_DEBUG_GPU_PARSER_LIMIT", "4")
    got, _, _ = run_pipeline(L, data, parser_threads=1, gops_per_window=2, gpu_parser=None)
    assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)
    p = L.Pipeline(data, gops_per_window=2, parser_threads=1, gpu_parser=None)
    try:
        assert p.info.gpu_parser == 0
        p.wait()
    finally:
        p.close()


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
