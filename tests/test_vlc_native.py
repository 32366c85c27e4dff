"""CPU: the native bitstream front end (libleon_vlc.so, include/leon_vlc.h) against
(a) the boundary tensors recorded from the UNMODIFIED reference parser (tests/golden/parser_*.json,
    made by tools/make_golden.js -- the reference drops B pictures, so those are covered by (b)),
(b) the product's JavaScript mirror of that parser under Node, picture by picture, B included,
(c) its own invariants: thread-count independence, sparse <-> dense round trip, key-map seek,
    and no crash on damaged input."""
import ctypes
import hashlib
import json
import os
import shutil
import subprocess

import ctypes as C

import numpy as np
import pytest

from helpers import ROOT, load_golden

import leon_vlc_ctypes as V

STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
NAMES = ["tiny_ip_32x32", "leon_synth_352x240", "ibbp_96x64", "slices5_ip_96x64"]
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def read(name):
    with open(os.path.join(STREAMS, name + ".jsv"), "rb") as f:
        return f.read()


def all_pictures(data, threads=0, dense=True):
    st = V.Stream(data, threads=threads)
    out = []
    while True:
        p = st.next_picture(dense=dense)
        if p is None:
            break
        out.append(p)
    return st, out


def test_library_exports_every_declared_symbol():
    lib = V.load()
    text = open(os.path.join(ROOT, "include", "leon_vlc.h")).read()
    for s in V.SYMBOLS:
        assert s + "(" in text, s
        assert getattr(lib, s) is not None


def _ref_uploads(pic):
    ups = [u for u in pic["uploads"] if "unit" in u]
    coef = [u["sha256"] for u in ups if u["unit"] == 0]
    unit2 = [u for u in ups if u["unit"] == 2]
    d = {"coef": coef, "qscale": unit2[0]["sha256"], "intra": [u for u in ups if u["unit"] == 4][0]["sha256"]}
    if pic["type"] == 2:
        d["repadd"] = unit2[1]["sha256"]
        d["mv"] = [u for u in ups if u["unit"] == 3][0]["sha256"]
    return d


@pytest.mark.parametrize("name", NAMES)
def test_tensors_equal_reference_parser(name):
    st, pics = all_pictures(read(name))
    ref = load_golden("parser_%s.json" % name)
    assert (st.info.mb_width, st.info.mb_height, st.info.coded_width) == (ref["mbWidth"], ref["mbHeight"], ref["codedWidth"])
    mine = [p for p in pics if p["type"] != 3]
    assert len(mine) == len(ref["pictures"]) > 0
    for i, (p, r) in enumerate(zip(mine, ref["pictures"])):
        assert p["type"] == r["type"], i
        u = _ref_uploads(r)
        assert [sha(p["coef_y"]), sha(p["coef_cb"]), sha(p["coef_cr"])] == u["coef"], "coefficients of picture %d" % i
        if name != "ibbp_96x64":      # see tests/test_js_parser.py: stale map entries differ once B pictures are read
            assert sha(p["qscale"]) == u["qscale"] and sha(p["intra"]) == u["intra"], i
            assert p["ts"] == r["ts"], i
        if p["type"] == 2:
            assert sha(p["repadd"]) == u["repadd"] and sha(p["mv_fwd"]) == u["mv"], i


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
@pytest.mark.parametrize("name", NAMES)
def test_tensors_equal_javascript_parser(name):
    cli = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js", "cli.js")
    out = subprocess.run(["node", cli, "tensors", os.path.join(STREAMS, name + ".jsv")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    js = json.loads(out.stdout)
    _, pics = all_pictures(read(name), threads=3)
    assert len(pics) == len(js["pictures"])
    for i, (p, j) in enumerate(zip(pics, js["pictures"])):
        assert p["type"] == j["type"], i
        keys = [("coefY", "coef_y"), ("coefCb", "coef_cb"), ("coefCr", "coef_cr"), ("qscale", "qscale"), ("intra", "intra")]
        if p["type"] != 1:
            keys += [("repadd", "repadd"), ("mvFwd", "mv_fwd")]
        if p["type"] == 3:
            keys += [("mvBwd", "mv_bwd"), ("mbDir", "mb_dir")]
        for a, b in keys:
            assert sha(p[b]) == j["sha"][a], (i, a)
        assert p["ts"] == j["ts"], i


@pytest.mark.parametrize("name", NAMES)
def test_result_does_not_depend_on_the_thread_count(name):
    _, one = all_pictures(read(name), threads=1, dense=False)
    _, many = all_pictures(read(name), threads=8, dense=False)
    assert len(one) == len(many)
    for a, b in zip(one, many):
        assert np.array_equal(a["grp_off"], b["grp_off"])
        # entries of a group come in no particular order
        for g in range(len(a["grp_off"]) - 1):
            lo, hi = a["grp_off"][g], a["grp_off"][g + 1]
            assert np.array_equal(np.sort(a["entries"][lo:hi]), np.sort(b["entries"][lo:hi]))
        for k in ("qscale", "intra", "repadd", "mv_fwd", "mv_bwd", "mb_dir"):
            assert (a[k] is None) == (b[k] is None) and (a[k] is None or np.array_equal(a[k], b[k])), k


def test_sparse_lists_round_trip_through_dense_planes():
    st, pics = all_pictures(read("leon_synth_352x240"))
    cw, ch = st.info.coded_width, st.info.coded_height
    for p in pics[:6]:
        grp_off, entries = V.sparsify(p["coef_y"], p["coef_cb"], p["coef_cr"], cw, ch)
        assert np.array_equal(grp_off, p["grp_off"])
        for g in range(len(grp_off) - 1):
            lo, hi = grp_off[g], grp_off[g + 1]
            assert np.array_equal(np.sort(entries[lo:hi]), np.sort(p["entries"][lo:hi]))
        nz = int(np.count_nonzero(p["coef_y"]) + np.count_nonzero(p["coef_cb"]) + np.count_nonzero(p["coef_cr"]))
        assert len(p["entries"]) == nz
        assert np.all(((p["entries"] >> 16) & 1) == 0) and np.all((p["entries"] >> 16) < 1024)


def test_key_map_seek_lands_on_the_second_gop():
    st = V.Stream(read("leon_synth_352x240"))
    assert st.info.keymap_count == 2
    assert st.seek(0.6) == 58896                  # = the JavaScript mirror, tests/test_js_parser.py
    n, first = 0, None
    while True:
        p = st.next_picture()
        if p is None:
            break
        first = p["type"] if first is None else first
        n += 1
    assert (n, first) == (12, 1)
    assert st.seek(0.0) == st.seek(-5.0)
    assert st.next_picture()["type"] == 1


def test_raw_elementary_stream_is_accepted():
    data = bytearray(read("tiny_ip_32x32"))
    i = data.find(b"\x00\x00\x01\xc3")
    es = bytes(data[i:]).replace(b"\x00\x00\x01\xc3", b"\x00\x00\x01\xb3")
    _, a = all_pictures(bytes(data))
    st, b = all_pictures(es)
    assert st.info.keymap_count == 0 and len(a) == len(b) == 3
    for x, y in zip(a, b):
        assert np.array_equal(x["coef_y"], y["coef_y"]) and np.array_equal(x["qscale"], y["qscale"])


def test_damaged_streams_fail_cleanly():
    lib = V.load()
    h = ctypes.c_void_p()
    assert lib.leon_vlc_open(b"\x00" * 4, 4, 1, ctypes.byref(h)) < 0
    assert lib.leon_vlc_open(b"\x12" * 64, 64, 1, ctypes.byref(h)) < 0 and b"sequence header" in lib.leon_vlc_last_error()
    data = read("leon_synth_352x240")
    rng = np.random.default_rng(5)
    for trial in range(40):
        d = bytearray(data)
        if trial % 2:
            d = d[: int(rng.integers(64, len(d)))]                     # truncation
        else:
            for _ in range(int(rng.integers(1, 40))):                  # bit flips behind the headers
                d[int(rng.integers(200, len(d)))] ^= 1 << int(rng.integers(0, 8))
        try:
            st = V.Stream(bytes(d), threads=2)
        except V.VlcError:
            continue
        for _ in range(40):
            try:
                if st.next_picture() is None:
                    break
            except V.VlcError:
                pass                      # an error is reported per picture; the parser moves on


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
@pytest.mark.parametrize("slice_mbs", [1, 4, 5, 16])
def test_slices_that_start_mid_row_and_span_rows(slice_mbs, tmp_path):
    """MPEG-1 slices need not be macroblock rows.  Streams whose slices are 1, 4, 5 or 16 macroblocks
    long (6 per row: mid-row starts, row-spanning slices) through the JavaScript mirror (serial slice
    loop of the reference) and through the native parser with 1 and 8 threads: same tensors."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import jsv_writer as W
    import synth as S
    rng = np.random.default_rng(100 + slice_mbs)
    cw, ch = 96, 64
    pics = []
    for ptype, disp, f, b in S.gop_ibbp(6):
        t = S.make_picture(rng, cw, ch, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
        t["display"] = disp
        pics.append(t)
    data, _ = W.write_stream(pics, cw, ch, cw, ch, gop_starts=[0], slice_mbs=slice_mbs)
    path = str(tmp_path / "s.jsv")
    with open(path, "wb") as fh:
        fh.write(data)
    cli = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js", "cli.js")
    out = subprocess.run(["node", cli, "tensors", path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    js = json.loads(out.stdout)["pictures"]
    for threads in (1, 8):
        _, mine = all_pictures(data, threads=threads)
        assert len(mine) == len(js) == len(pics)
        for i, (p, j, t) in enumerate(zip(mine, js, pics)):
            assert p["n_slices"] == -(-24 // slice_mbs)
            assert sha(p["coef_y"]) == j["sha"]["coefY"] == sha(t["coef_y"].astype("<i2")), (threads, i)
            assert sha(p["coef_cb"]) == j["sha"]["coefCb"] and sha(p["coef_cr"]) == j["sha"]["coefCr"], (threads, i)
            assert sha(p["qscale"]) == j["sha"]["qscale"] and sha(p["intra"]) == j["sha"]["intra"], (threads, i)
            if p["type"] != 1:
                assert sha(p["mv_fwd"]) == j["sha"]["mvFwd"] and sha(p["repadd"]) == j["sha"]["repadd"], (threads, i)
            if p["type"] == 3:
                assert sha(p["mv_bwd"]) == j["sha"]["mvBwd"] and sha(p["mb_dir"]) == j["sha"]["mbDir"], (threads, i)


@pytest.mark.parametrize("name", ["leon_synth_352x240", "ibbp_96x64"])
def test_gop_shards_parse_like_the_whole_stream(name):
    """a stream opened on the bytes of ONE key-map entry (a GOP shard, starting with its 00 00 01 C3
    sequence header) gives exactly that GOP's pictures -- the unit the pipeline's parser threads and the
    multi-GPU partition work on (decoders/jsv.js:264-350)"""
    data = read(name)
    whole_st, whole = all_pictures(data, threads=1)
    offs = whole_st.keymap()
    assert len(offs) == whole_st.info.keymap_count >= 2
    got = []
    for b, e in whole_st.shard_ranges():
        assert data[b:b + 4] == b"\x00\x00\x01\xc3"
        _, pics = all_pictures(data[b:e], threads=1)
        assert pics and pics[0]["type"] == 1
        got += pics
    assert len(got) == len(whole)
    for i, (a, b) in enumerate(zip(got, whole)):
        assert a["type"] == b["type"] and a["temporal_reference"] == b["temporal_reference"], i
        for k in ("coef_y", "coef_cb", "coef_cr", "repadd", "mv_fwd", "mv_bwd", "mb_dir"):
            if b.get(k) is not None:
                assert np.array_equal(a[k], b[k]), (i, k)


@pytest.mark.parametrize("name", ["leon_synth_352x240", "slices5_ip_96x64", "ibbp_96x64", "yuva_ibbp_96x64"])
def test_scan_picture_finds_what_the_parser_parses(name):
    """leon_vlc_scan_picture (the picture layer only, for the GPU parser): the same pictures in the same order as the
    full parse, as many slices, every position right behind a slice start code whose code byte is the slice's"""
    data = read(name)
    _, whole = all_pictures(data, threads=1)
    st = V.Stream(data, threads=1)
    n = 0
    while True:
        sc = st.scan_picture()
        if sc is None:
            break
        full = whole[n]
        assert (sc["type"], sc["temporal_reference"]) == (full["type"], full["temporal_reference"])
        assert len(sc["slice_code"]) == full["n_slices"] > 0
        for code, pos in zip(sc["slice_code"], sc["slice_bit_pos"]):
            assert pos % 8 == 0 and data[pos // 8 - 4:pos // 8] == bytes([0, 0, 1, code])
        assert sc["slice_bit_pos"] == sorted(sc["slice_bit_pos"]) and sc["end_byte"] * 8 >= sc["slice_bit_pos"][-1]
        assert data[sc["end_byte"]:sc["end_byte"] + 3] in (b"\x00\x00\x01", b"") or sc["end_byte"] >= len(data) - 3
        n += 1
    assert n == len(whole)


@pytest.mark.parametrize("name", ["ibbp_96x64", "slices5_ip_96x64", "yuva_ibbp_96x64"])
def test_scan_stream_reads_in_place_what_a_copied_stream_reads(name):
    """leon_vlc_open_scan (no copy, no read-ahead thread; the pipeline's gpu_parser mode opens every GOP shard with it):
    the same scan results as a stream opened the ordinary way; the parsing calls are refused on it"""
    data = read(name)
    a, b = V.Stream(data, threads=1), V.Stream(data, scan_only=True)
    assert (b.info.coded_width, b.info.coded_height, b.info.keymap_count) == (a.info.coded_width, a.info.coded_height, a.info.keymap_count)
    n = 0
    while True:
        x, y = a.scan_picture(), b.scan_picture()
        assert x == y
        if x is None:
            break
        n += 1
    assert n > 0
    c = V.Stream(data, scan_only=True)
    with pytest.raises(V.VlcError, match="scan_picture only"):
        c.next_picture()


def test_gpu_tables_are_the_parsers_tables():
    """leon_vlc_get_gpu_tables: every entry decodes back to a code of the right length; the 12-bit coefficient table
    agrees with the 16-bit one wherever it answers"""
    t = V.GpuTables()
    lib = V.load()
    lib.leon_vlc_get_gpu_tables.argtypes = [C.POINTER(V.GpuTables)]
    assert lib.leon_vlc_get_gpu_tables(C.byref(t)) == 0
    fast, coef = np.array(t.fast12), np.array(t.coef16)
    assert (fast[(np.arange(4096) >> 10) == 2] == (0x80 | 2)).all()                 # '10': end of block
    for p in range(4096):
        f = int(fast[p])
        if f == 0 or (f & 0x80) or (p >> 10) == 3:
            continue
        e = int(coef[p << 4])
        ln, cf = e >> 16, e & 0xffff
        assert ln + 1 == (f & 0x7f) and (cf >> 8) == ((f >> 8) & 0xff), p
        level = cf & 0xff
        got = (f >> 16) - 65536 if (f >> 16) >= 32768 else (f >> 16)
        assert got == (-level if (p >> (11 - ln)) & 1 else level), p
    zz = np.array(t.zz_off)
    assert sorted(zz.tolist()) == sorted(r * 128 + c * 2 for r in range(8) for c in range(8))
    assert all(1 <= (int(e) & 0xffff) <= 35 and 1 <= (int(e) >> 16) <= 11 for e in np.array(t.mba) if e)


def test_a_shard_keeps_its_last_macroblock():
    """a GOP whose last picture ends in a macroblock of two bytes (vectors, no coefficients): cut at the key-map
    offset itself the slice loop would take those bytes for the end of the data (jsv.js:1710-1760) and drop the
    macroblock; the shard therefore takes the 00 00 01 of what follows along"""
    import jsv_writer as W
    import synth as S
    rng = np.random.default_rng(26)
    pics, starts = [], []
    for n in (6, 9, 3):
        starts.append(len(pics))
        for ptype, disp, f, b in S.gop_ibbp(n):
            t = S.make_picture(rng, 208, 112, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            t["display"] = disp
            pics.append(t)
    data = W.write_stream(pics, 208, 112, 208, 112, gop_starts=starts)[0]
    whole_st, whole = all_pictures(data, threads=1)
    assert whole[5]["type"] == 3 and whole[5]["mb_dir"][-1] != 0          # the picture and the macroblock this is about
    got = []
    for b, e in whole_st.shard_ranges():
        got += all_pictures(data[b:e], threads=1)[1]
    assert len(got) == len(whole)
    for i, (a, b) in enumerate(zip(got, whole)):
        for k in ("coef_y", "coef_cb", "coef_cr", "repadd", "mv_fwd", "mv_bwd", "mb_dir"):
            if b.get(k) is not None:
                assert np.array_equal(a[k], b[k]), (i, k)
    # ... and cut at the offset itself it is lost
    offs = whole_st.keymap()
    short = all_pictures(data[offs[0]:offs[1]], threads=1)[1]
    assert short[5]["mb_dir"][-1] == 0


def test_corruption_fuzz_under_address_and_ub_sanitizers(tmp_path):
    """the front end built with -fsanitize=address,undefined parses randomly damaged and truncated copies of the
    fixture streams (yuva and B pictures included): every case ends in a clean refusal or a clean parse"""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "vlc_fuzz")
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-pthread",
                        "-o", exe, os.path.join(ROOT, "tools", "vlc_fuzz.cpp"),
                        os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "csrc", "leon_vlc.cpp")], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    streams = [os.path.join(STREAMS, n + ".jsv") for n in ("leon_synth_352x240", "yuva_ibbp_96x64", "ibbp_96x64", "slices5_ip_96x64")]
    r = subprocess.run([exe] + streams, capture_output=True, text=True, timeout=600, env=dict(os.environ, LEON_FUZZ_CASES="150"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "refused" in r.stdout


def test_merged_gops_parse_like_the_parts():
    """jsv_writer.merge_gops: GOPs written as separate single-GOP streams (by parallel processes, for the long 1080p
    stream of tools/stream_1080p.py) and merged parse to the same pictures as the parts, the key map points at each GOP's
    sequence header and the GOP time codes count on"""
    import jsv_writer as W
    import synth as S
    cw, ch = 96, 64
    parts, want = [], []
    for g, n in enumerate([6, 9, 3]):
        rng = np.random.default_rng([5, g])
        pics = []
        for ptype, disp, f, b in S.gop_ibbp(n):
            t = S.make_picture(rng, cw, ch, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            t["display"] = disp
            pics.append(t)
        data = W.write_stream(pics, cw, ch, cw, ch, gop_starts=[0])[0]
        parts.append(data)
        st = V.Stream(data, threads=1)
        while True:
            p = st.next_picture(dense=True)
            if p is None:
                break
            want.append(p)
        st.close()
    merged, offs = W.merge_gops(parts, cw, ch)
    st = V.Stream(merged, threads=1)
    assert list(st.keymap()) == offs and len(offs) == 3
    got = []
    while True:
        p = st.next_picture(dense=True)
        if p is None:
            break
        got.append(p)
    assert len(got) == len(want) == 18
    for a, b in zip(got, want):
        assert a["type"] == b["type"] and a["temporal_reference"] == b["temporal_reference"]
        for k in ("coef_y", "coef_cb", "coef_cr", "qscale", "intra", "repadd", "mv_fwd", "mv_bwd", "mb_dir"):
            if b.get(k) is not None:
                assert np.array_equal(a[k], b[k]), k
    # time codes: GOP 1 starts 6 pictures in, GOP 2 15 pictures in (25 pictures/s)
    firsts = [p["ts"] for p in got if p["type"] == 1]
    assert firsts[0] < firsts[1] < firsts[2] and abs((firsts[1] - firsts[0]) - 6 * 40.0) < 1e-6 and abs((firsts[2] - firsts[0]) - 15 * 40.0) < 1e-6
