"""CPU: the product's JavaScript bitstream layer (mpeg1video-decoder-webgl_amd/js) under Node
against (a) the boundary tensors recorded from the UNMODIFIED reference parser
(tests/golden/parser_*.json, made by tools/make_golden.js) and (b) the tensors the stream
writer encoded (B pictures, which the reference parser drops)."""
import base64
import contextlib
import hashlib
import io
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ROOT, load_golden

JS = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js", "cli.js")
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")


def run_cli(*args):
    out = subprocess.run(["node", JS] + list(args), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout)


def _ref_uploads(pic):
    ups = [u for u in pic["uploads"] if "unit" in u]
    coef = [u["sha256"] for u in ups if u["unit"] == 0]
    unit2 = [u for u in ups if u["unit"] == 2]
    d = {"coef": coef, "qscale": unit2[0]["sha256"], "intra": [u for u in ups if u["unit"] == 4][0]["sha256"]}
    if pic["type"] == 2:
        d["repadd"] = unit2[1]["sha256"]
        d["mv"] = [u for u in ups if u["unit"] == 3][0]["sha256"]
    return d


@pytest.mark.parametrize("name", ["tiny_ip_32x32", "leon_synth_352x240", "ibbp_96x64", "slices5_ip_96x64"])
def test_tensors_equal_reference_parser(name):
    got = run_cli("tensors", os.path.join(STREAMS, name + ".jsv"))
    ref = load_golden("parser_%s.json" % name)
    assert (got["mbWidth"], got["mbHeight"], got["codedWidth"]) == (ref["mbWidth"], ref["mbHeight"], ref["codedWidth"])
    mine = [p for p in got["pictures"] if p["type"] != 3]          # the reference drops B pictures
    assert len(mine) == len(ref["pictures"]) > 0
    for i, (p, r) in enumerate(zip(mine, ref["pictures"])):
        assert p["type"] == r["type"], i
        u = _ref_uploads(r)
        assert [p["sha"]["coefY"], p["sha"]["coefCb"], p["sha"]["coefCr"]] == u["coef"], "coefficients of picture %d" % i
        if name != "ibbp_96x64":
            # the persistent maps keep stale entries for skipped macroblocks; once B pictures are
            # decoded in between (which the reference drops) the stale entries legitimately differ
            assert p["sha"]["qscale"] == u["qscale"] and p["sha"]["intra"] == u["intra"], "maps of picture %d" % i
        if p["type"] == 2:
            assert p["sha"]["repadd"] == u["repadd"] and p["sha"]["mvFwd"] == u["mv"], "P maps of picture %d" % i
        if name != "ibbp_96x64":      # with B pictures dropped the reference's 'ts' bookkeeping differs
            assert p["ts"] == r["ts"], i
    if name != "ibbp_96x64":
        assert [e["ev"] for e in got["events"]] == [e["ev"] for e in ref["events"]]


def test_b_pictures_equal_what_the_writer_encoded():
    import make_streams as M
    import synth as S
    with contextlib.redirect_stdout(io.StringIO()):
        tmp_root = M.ROOT
        pics = None
        # regenerate the same tensors (deterministic seed) without touching the fixture file
        rng = np.random.default_rng(7)
        pics = []
        for gop in (S.gop_ibbp(12), S.gop_ibbp(6)):
            for ptype, disp, f, b in gop:
                force = 2 if (ptype == S.PIC_B and f is None) else None
                t = S.make_picture(rng, 96, 64, ptype, force_dir=force)
                pics.append(t)
        del tmp_root
    got = run_cli("tensors", os.path.join(STREAMS, "ibbp_96x64.jsv"))
    assert len(got["pictures"]) == len(pics) == 18
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    d64 = lambda s, dt: np.frombuffer(base64.b64decode(s), dtype=dt)
    for t, p in zip(pics, got["pictures"]):
        assert p["type"] == t["type"]
        assert p["sha"]["coefY"] == sha(t["coef_y"].astype("<i2")) and p["sha"]["coefCb"] == sha(t["coef_cb"].astype("<i2"))
        if t["type"] == 3:
            ni = t["intra"] == 0
            d = d64(p["mbDir"], np.uint8)
            assert np.array_equal(d[ni], t["mb_dir"][ni])
            mf = d64(p["mvFwd"], "<i2").reshape(-1, 2)
            mb = d64(p["mvBwd"], "<i2").reshape(-1, 2)
            fw = ni & ((t["mb_dir"] & 1) != 0)
            bw = ni & ((t["mb_dir"] & 2) != 0)
            assert np.array_equal(mf[fw], t["mv_fwd"].reshape(-1, 2)[fw])
            assert np.array_equal(mb[bw], t["mv_bwd"].reshape(-1, 2)[bw])
            assert np.array_equal(d64(p["repadd"], np.uint8), t["repadd"])


def test_key_map_seek():
    got = run_cli("tensors", os.path.join(STREAMS, "leon_synth_352x240.jsv"))
    assert len(got["pictures"]) == 24
    # key entry 1 carries timecode frame 12 -> (12+1)/25 = 0.52 s (jsv.js:315-325): seek(0.6) lands on it
    out = subprocess.run(["node", "-e", """
      const {JsvDecoder} = require(%r); const fs = require('fs');
      const d = new JsvDecoder({}); let n = 0, first = -1;
      d.on('frame', f => { if (first < 0) first = f.type; n++; });
      d.addBuffer(new Uint8Array(fs.readFileSync(%r))); d._initMeta();
      const off = d.seek(0.6); while (d.decodeFrame());
      console.log(JSON.stringify({n, first, off, count: d._keyMap.count}));
    """ % (os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js", "jsv_decoder.js"),
           os.path.join(STREAMS, "leon_synth_352x240.jsv"))], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r == {"n": 12, "first": 1, "off": 58896, "count": 2}


@pytest.mark.parametrize("name", ["tiny_ip_32x32", "leon_synth_352x240", "ibbp_96x64", "slices5_ip_96x64"])
def test_native_front_end_under_node_equals_the_javascript_parser(name):
    """js/native_decoder.js (libleon_vlc through napi/leon_vlc_napi.node, sparse lists densified)
    against js/jsv_decoder.js: same tensors, same time stamps, same events, picture by picture."""
    a = run_cli("tensors", os.path.join(STREAMS, name + ".jsv"))
    b = run_cli("tensors", os.path.join(STREAMS, name + ".jsv"), "--native")
    assert len(a["pictures"]) == len(b["pictures"]) > 0
    for i, (x, y) in enumerate(zip(a["pictures"], b["pictures"])):
        assert (x["type"], x["ts"], x["temporalReference"]) == (y["type"], y["ts"], y["temporalReference"]), i
        assert x["sha"] == y["sha"], i
    assert a["events"] == b["events"]
    assert (a["codedWidth"], a["codedHeight"], a["mbWidth"], a["mbHeight"]) == (b["codedWidth"], b["codedHeight"], b["mbWidth"], b["mbHeight"])


def test_native_front_end_seek_equals_the_javascript_parser():
    a = run_cli("tensors", os.path.join(STREAMS, "leon_synth_352x240.jsv"), "--seek=0.6")
    b = run_cli("tensors", os.path.join(STREAMS, "leon_synth_352x240.jsv"), "--seek=0.6", "--native")
    assert [e for e in a["events"] if e["ev"] == "seeked"] == [e for e in b["events"] if e["ev"] == "seeked"] != []
    assert len(a["pictures"]) == len(b["pictures"]) == 12
    assert [p["sha"] for p in a["pictures"]] == [p["sha"] for p in b["pictures"]]


@pytest.mark.parametrize("name", ["tiny_ip_32x32", "leon_synth_352x240", "ibbp_96x64"])
def test_es_jsv_transcoder_round_trip(name, tmp_path):
    """js/es2jsv.js: JSV -> elementary stream (header and key map dropped, C3 -> B3) -> JSV rebuilds the
    fixture byte for byte (header, duration, key-map offsets and time codes), and the elementary stream
    parses to the same pictures."""
    tool = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js", "es2jsv.js")
    src = os.path.join(STREAMS, name + ".jsv")
    es, back = str(tmp_path / "a.m1v"), str(tmp_path / "b.jsv")
    for args in (["to-es", src, es], ["to-jsv", es, back]):
        out = subprocess.run(["node", tool] + args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
    assert open(back, "rb").read() == open(src, "rb").read()
    raw = open(es, "rb").read()
    assert raw[:4] == b"\x00\x00\x01\xb3" and b"\x00\x00\x01\xc3" not in raw
    import leon_vlc_ctypes as V
    a, b = V.Stream(open(src, "rb").read(), threads=2), V.Stream(raw, threads=2)
    n = 0
    while True:
        p, q = a.next_picture(), b.next_picture()
        assert (p is None) == (q is None)
        if p is None:
            break
        n += 1
        assert p["type"] == q["type"] and np.array_equal(p["grp_off"], q["grp_off"]) and np.array_equal(p["qscale"], q["qscale"])
        assert np.array_equal(np.sort(p["entries"]), np.sort(q["entries"]))
    assert n > 0
