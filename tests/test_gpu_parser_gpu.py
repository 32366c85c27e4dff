"""GPU: the slice layer decoded on the GPU (leon_pipeline_config.gpu_parser; csrc/leon_vlc_gpu.h: one lane per
slice, decodeSlice .. decodeBlockGL of decoders/jsv.js:683-1525 as csrc/leon_vlc.cpp reads them) -- the pipeline's
frames against the oracle run on the host front end's tensors, and against the pipeline with the host front end,
on every fixture stream (I/P/B, several slices per row, custom matrices, yuva), on multi-GOP streams in several
window / ring geometries, and on damaged streams."""
import os

import numpy as np
import pytest

from helpers import ROOT
from test_pipeline_gpu import STREAMS, ibbp_stream, oracle_frames, run_pipeline

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import leon_ctypes
    leon_ctypes.load()
    return leon_ctypes


@pytest.mark.parametrize("name", ["leon_synth_352x240", "slices5_ip_96x64", "custom_intra_ip_48x32", "tiny_ip_32x32", "yuva_ibbp_96x64"])
def test_fixture_streams(L, name):
    data = open(os.path.join(STREAMS, name + ".jsv"), "rb").read()
    want = oracle_frames(data)
    got, order, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=2, gpu_parser=True)
    assert set(got) == set(want) and stats["pictures"] == len(want)
    for k in want:
        bad = np.argwhere(got[k] != want[k])
        assert bad.size == 0, "%s: frame %s differs in %d bytes, first at %s" % (name, k, len(bad), bad[0])
    assert order == sorted(order)


@pytest.mark.parametrize("window,threads,inflight", [(1, 1, 1), (3, 4, 2), (8, 2, 3)])
def test_ibbp_windows_and_loops(L, window, threads, inflight):
    data = ibbp_stream(208, 112, [6, 9, 3, 12, 6, 9, 12], seed=23 + window)
    want = oracle_frames(data)
    got, order, stats = run_pipeline(L, data, parser_threads=threads, gops_per_window=window, windows_in_flight=inflight, loop=2,
                                     gpu_parser=True)
    assert stats["pictures"] == 2 * len(want)
    for (g, k), rgba in sorted(got.items()):
        bad = np.argwhere((rgba != want[(g % 7, k)]).any(axis=2))
        assert bad.size == 0, "GOP %d frame %d: %d pixels differ, rows %d..%d, columns %d..%d" % (
            g, k, len(bad), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max())


def test_same_frames_as_the_host_front_end_at_1080p(L):
    """two 1080p GOPs (68 slices per picture, tools/stream_1080p.py writes the stream when it is not there): byte for
    byte the frames of the pipeline that parses on the host -- and both equal the oracle run on the host front end's
    tensors, all 24 pictures"""
    import stream_1080p
    data = stream_1080p.load()
    host, _, _ = run_pipeline(L, data, parser_threads=8, gops_per_window=2)
    gpu, _, _ = run_pipeline(L, data, parser_threads=2, gops_per_window=2, gpu_parser=True)
    assert set(host) == set(gpu) and len(gpu) == 24
    for k in host:
        assert np.array_equal(host[k], gpu[k]), k
    want = oracle_frames(data)
    assert set(want) == set(gpu)
    for k in sorted(want):
        bad = np.argwhere((gpu[k] != want[k]).any(axis=2))
        assert bad.size == 0, "frame %s: %d pixels differ from the oracle, rows %d..%d" % (k, len(bad), bad[:, 0].min(), bad[:, 0].max())


def test_sixteen_different_1080p_gops_same_frames_from_both_front_ends(L):
    """the stream the end-to-end figures are quoted on (tools/stream_1080p.py ensure_varied: 16 closed IBBP GOPs of different
    content, quantiser scales 2 .. 31 by row, 192 pictures): every frame of the GPU-parsed pipeline equals the host-parsed
    one (SHA-256 per frame), and two of its GOPs equal the ORACLE's frames too (two product paths agreeing with
    each other is not parity; the two-GOP stream above is checked against the oracle in full)"""
    import hashlib
    import stream_1080p
    data = stream_1080p.load_varied()

    def digests(gpu_parser):
        out = {}

        def on_window(window, frames):
            for f in frames:
                out[(f["gop"], f["display_index"])] = hashlib.sha256(L.read_frame(f).tobytes()).hexdigest()
        pipe = L.Pipeline(data, parser_threads=8, gops_per_window=8, gpu_parser=gpu_parser, on_window=on_window)
        try:
            pipe.wait()
            assert pipe.ended and pipe.error is None
        finally:
            pipe.close()
        return out
    gpu, host = digests(True), digests(False)
    assert len(gpu) == 16 * 12 and sorted(gpu) == sorted(host)
    bad = [k for k in sorted(gpu) if gpu[k] != host[k]]
    assert not bad, bad[:8]
    import bench
    want = bench.oracle_stream_sums(data, [3, 11], threads=2, digest=lambda rgba: hashlib.sha256(np.ascontiguousarray(rgba).tobytes()).hexdigest())
    assert len(want) == 24
    bad = [k for k in sorted(want) if gpu[k] != want[k]]
    assert not bad, "frames that differ from the oracle: %s" % bad[:8]


def test_damaged_streams_are_refused(L):
    """bytes of one GOP overwritten: the run ends with an error instead of delivering garbage -- whichever layer
    notices (the host's picture layer, or a slice on the GPU when its window completes)"""
    import leon_vlc_ctypes as V
    good = ibbp_stream(96, 64, [6, 6, 6], seed=9)
    offs = V.Stream(good, threads=1).keymap()
    for lo, hi, fill in ((60, 400, 0xFF), (200, 260, 0x00), (120, 121, 0x5A)):
        bad = bytearray(good)
        for i in range(offs[1] + lo, offs[1] + hi):
            bad[i] = fill
        pipe = L.Pipeline(bytes(bad), gops_per_window=1, parser_threads=1, max_gop_pictures=64, gpu_parser=True)
        try:
            try:
                pipe.wait()
                clean = True
            except L.LeonError:
                clean = False
            # a single flipped byte may still be a valid stream; the long runs are not
            assert not clean or hi - lo == 1
        finally:
            pipe.close()


def test_random_corruption_never_hangs_or_faults(L):
    """40 seeded corruptions of a three-GOP stream (byte runs overwritten with random bytes, zeros or ones, inside the
    second GOP's slice data): every run ends -- with frames or with an error -- and the next one starts on a clean
    device; a run that completes delivers every frame of the untouched GOPs intact"""
    import leon_vlc_ctypes as V
    good = ibbp_stream(96, 64, [6, 6, 6], seed=9)
    offs = V.Stream(good, threads=1).keymap()
    clean, _, _ = run_pipeline(L, good, parser_threads=1, gops_per_window=3, gpu_parser=True, max_gop_pictures=64)
    rng = np.random.default_rng(2026)
    refused = 0
    for case in range(40):
        bad = bytearray(good)
        lo = int(rng.integers(offs[1] + 40, offs[2] - 40))
        n = int(rng.integers(1, 48))
        kind = case % 3
        for i in range(lo, min(lo + n, offs[2] - 8)):
            bad[i] = int(rng.integers(0, 256)) if kind == 0 else (0 if kind == 1 else 0xFF)
        got = {}

        def on_window(window, frames):
            for f in frames:
                got[(f["gop"], f["display_index"])] = L.read_frame(f)
        pipe = L.Pipeline(bytes(bad), gops_per_window=3, parser_threads=1, max_gop_pictures=64, gpu_parser=True, on_window=on_window)
        try:
            try:
                pipe.wait()
                for k, rgba in got.items():
                    if k[0] != 1:
                        assert np.array_equal(rgba, clean[k]), (case, k)
            except L.LeonError:
                refused += 1
        finally:
            pipe.close()
    assert refused > 0


def test_a_picture_whose_group_counters_need_more_than_64_kb_of_lds(L):
    """k_vlc_index counts a picture's groups in dynamic LDS, four bytes each: 3584 x 2048 has 21504 groups = 86 KB, beyond
    what a launch gets without hipFuncAttributeMaxDynamicSharedMemorySize (the pipeline asks for it at create).  One
    I B B P B B GOP, GPU-parsed frames against the host-parsed pipeline, and the I and P picture against the oracle."""
    data = ibbp_stream(3584, 2048, [6], seed=77)
    gpu, order, stats = run_pipeline(L, data, parser_threads=1, gops_per_window=1, gpu_parser=True)
    host, _, _ = run_pipeline(L, data, parser_threads=1, gops_per_window=1)
    assert stats["pictures"] == 6 and sorted(gpu) == sorted(host) == [(0, d) for d in range(6)]
    for k in sorted(gpu):
        assert np.array_equal(gpu[k], host[k]), k
    want = oracle_frames(data)
    for k in ((0, 2), (0, 5), (0, 3)):
        assert np.array_equal(gpu[k].reshape(-1), np.asarray(want[k]).reshape(-1)), k


def _stream_of_shortest_codes(cw, ch):
    """I B B P B B (coded order I B B P B B of S.gop_ibbp) whose every coefficient is +-1 at every position: a coded block is 64 symbols of three bits ('11s';
    '1s' in first position) -- the densest stream of entries the syntax allows: what the GPU parser's arenas are sized for
    (an entry per three bits; csrc/leon_pipeline_impl.h scan_gop_for_gpu)"""
    import jsv_writer as W
    import synth as S
    rng = np.random.default_rng(77)
    pics = []
    for ptype, disp, f, b in S.gop_ibbp(6):
        t = S.make_picture(rng, cw, ch, ptype, intra_frac=0.05, skip_frac=0.0, uncoded_frac=0.0,
                           force_dir=2 if (ptype == S.PIC_B and f is None) else None)
        for name in ("coef_y", "coef_cb", "coef_cr"):
            sign = rng.integers(0, 2, size=t[name].shape) * 2 - 1
            t[name] = sign.astype(t[name].dtype)
        t["display"] = disp
        pics.append(t)
    return W.write_stream(pics, cw, ch, cw, ch, gop_starts=[0])[0]


def test_the_densest_stream_the_syntax_allows_fits_the_entry_lists(L):
    data = _stream_of_shortest_codes(96, 64)
    want = oracle_frames(data)
    host, _, hstats = run_pipeline(L, data, parser_threads=1, gops_per_window=1)
    # the premise: close to an entry per three bits of stream (container, headers and macroblock layers included)
    assert hstats["entries"] * 3.0 / (8 * len(data)) > 0.9, (hstats["entries"], len(data))
    gpu, _, _ = run_pipeline(L, data, parser_threads=1, gops_per_window=1, gpu_parser=True)
    assert set(gpu) == set(want) == set(host)
    for k in want:
        assert np.array_equal(host[k], want[k]), k
        assert np.array_equal(gpu[k], want[k]), k


def test_overlapping_slices_are_refused_by_the_gpu_parser_and_decoded_by_the_host_parser(L):
    """MPEG-1 forbids slices that overlap; the reference (and the host parser) decode them one after the other, the later
    one wins.  On the GPU the slices of a picture are decoded side by side and write their block records side by side
    (leon_vlc_gpu.h VlcSliceOut): such a picture is refused, deterministically, instead of decoded in an order nobody knows.
    The stream: the second slice of the first picture renamed to row 1."""
    good = ibbp_stream(96, 64, [3, 3], seed=5)
    at = good.find(b"\x00\x00\x01\x02")
    assert at > 0 and good[:at].count(b"\x00\x00\x01\x01") == 1
    bad = bytearray(good)
    bad[at + 3] = 1
    bad = bytes(bad)
    want = oracle_frames(bad)                    # the host front end + the oracle: the renamed slice overwrites row 1, row 2 stays empty
    host, _, _ = run_pipeline(L, bad, parser_threads=1, gops_per_window=2)
    assert set(host) == set(want)
    for k in want:
        assert np.array_equal(host[k], want[k]), k
    pipe = L.Pipeline(bad, gops_per_window=2, parser_threads=1, gpu_parser=True)
    try:
        with pytest.raises(L.LeonError, match="overlap"):
            pipe.wait()
    finally:
        pipe.close()
    # and the untouched stream decodes
    gpu, _, _ = run_pipeline(L, good, parser_threads=1, gops_per_window=2, gpu_parser=True)
    want = oracle_frames(good)
    for k in want:
        assert np.array_equal(gpu[k], want[k]), k
