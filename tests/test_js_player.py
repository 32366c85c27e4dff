"""The headless HTML5-video-shaped facade (js/leon_player.js): event order, display re-ordering of
B pictures, queue bound and seek -- bitstream-only on CPU, full pixels on the GPU."""
import json
import os
import shutil
import subprocess

import pytest

from helpers import ROOT

JSDIR = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "js")
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")

_SCRIPT = r"""
const path = require('path');
const {LeonPlayer} = require(path.join(%(js)r, 'leon_player.js'));
const crypto = require('crypto');
const backend = %(gpu)s ? require(path.join(%(js)r, '..', 'napi', 'leon_napi.node')) : null;
const ev = [], shown = [];
let maxQueue = 0;
const p = new LeonPlayer({backend, realtime: false, nativeParser: %(native)s,
  render: (rgba, f) => shown[shown.length - 1].rgba = crypto.createHash('sha256').update(Buffer.from(rgba.buffer, rgba.byteOffset, rgba.byteLength)).digest('hex')});
for (const n of ['loadstart','loadedmetadata','loadeddata','canplay','canplaythrough','play','playing','pause','seeking','seeked','ended','error'])
  p.on(n, () => ev.push(n));
p.on('timeupdate', (e) => { maxQueue = Math.max(maxQueue, p._decodedFrames.length); });
const orig = p._displayFrame.bind(p);
p._displayFrame = function () { const f = p._decodedFrames[0]; if (f) shown.push({type: f.type, tr: f.temporalReference, index: f.index}); orig(); };
p.src = %(stream)r;
const meta = {duration: p.duration, w: p.videoWidth, h: p.videoHeight, rs: p.readyState};
p.play();
const first = {ended: p.ended, paused: p.paused, n: p.framesDisplayed, t: p.currentTime};
%(extra)s
p.destroy();
console.log(JSON.stringify({ev, shown, meta, first, maxQueue, canPlay: [p.canPlayType('video/jsv'), p.canPlayType('video/mp4')]}));
"""


def _run(stream, gpu, extra="", native=False):
    src = _SCRIPT % {"js": JSDIR, "gpu": "true" if gpu else "false", "stream": os.path.join(STREAMS, stream), "extra": extra,
                     "native": "true" if native else "false"}
    out = subprocess.run(["node", "-e", src], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout)


def test_events_and_display_order_bitstream_only():
    r = _run("ibbp_96x64.jsv", gpu=False)
    assert r["ev"][:5] == ["loadstart", "loadedmetadata", "loadeddata", "canplay", "canplaythrough"]
    assert r["ev"][5:7] == ["play", "playing"] and r["ev"][-1] == "ended"
    assert r["meta"]["w"] == 90 and r["meta"]["h"] == 60 and r["meta"]["rs"] == 4
    assert r["first"] == {"ended": True, "paused": True, "n": 18, "t": r["first"]["t"]}
    # display order = temporal reference order inside each GOP (12 + 6 pictures)
    trs = [s["tr"] for s in r["shown"]]
    assert trs == list(range(12)) + list(range(6))
    assert r["maxQueue"] <= 10                      # MAX_DECODED_FRAMES (player/parts/end.js:57)
    assert r["canPlay"] == ["probably", ""]


def test_native_parser_option_gives_the_same_playback():
    a = _run("ibbp_96x64.jsv", gpu=False)
    b = _run("ibbp_96x64.jsv", gpu=False, native=True)
    assert a == b


@pytest.mark.parametrize("native", [False, True])
def test_seek_restarts_at_the_key_entry(native):
    extra = "p.currentTime = 0.6; p.play(); first.afterSeek = p.framesDisplayed;"
    r = _run("leon_synth_352x240.jsv", gpu=False, extra=extra, native=native)
    assert r["first"]["n"] == 24 and r["first"]["afterSeek"] == 24 + 12
    assert "seeking" in r["ev"] and "seeked" in r["ev"] and r["ev"].count("ended") == 2


@pytest.mark.gpu
def test_player_pixels_match_decode_order_output():
    """Every displayed frame's RGBA equals the frame the plain decode loop produced for it."""
    from test_js_parser import run_cli
    r = _run("ibbp_96x64.jsv", gpu=True)
    dec = run_cli("decode", os.path.join(STREAMS, "ibbp_96x64.jsv"), "--rgba")
    by_index = {i: p["rgba"] for i, p in enumerate(dec["pictures"])}
    assert len(r["shown"]) == 18 and r["ev"][-1] == "ended"
    for s in r["shown"]:
        assert s["rgba"] == by_index[s["index"]], s
    # the same playback with the native front end and the sparse boundary
    assert _run("ibbp_96x64.jsv", gpu=True, native=True)["shown"] == r["shown"]


_PIPE_SCRIPT = r"""
const path = require('path'), crypto = require('crypto');
const { LeonPlayer } = require(path.join(%(js)r, 'leon_player.js'));
const backend = require(path.join(%(js)r, '..', 'napi', 'leon_napi.node'));
const ev = [], shown = [];
const p = new LeonPlayer({ backend, pipeline: true, realtime: false, parserThreads: 2, gpuParser: %(gpu_parser)s,
  render: (rgba, f) => shown.push({ gop: f.gop, di: f.displayIndex, ts: f.ts, sha: crypto.createHash('sha256').update(rgba).digest('hex') }) });
for (const e of ['loadstart', 'loadedmetadata', 'loadeddata', 'canplay', 'play', 'playing', 'seeking', 'seeked', 'ended', 'error']) p.on(e, () => ev.push(e));
let phase = 0;
p.on('ended', () => {
  if (phase === 0) { phase = 1; p.currentTime = %(seek)s; p.play(); return; }
  console.log(JSON.stringify({ ev, shown, w: p.videoWidth, h: p.videoHeight, n: p.framesDisplayed }));
  p.destroy();
});
p.src = %(stream)r;
p.play();
"""


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_parser", [False, True], ids=["host-parser", "gpu-parser"])
def test_player_over_the_native_pipeline(gpu_parser):
    """the HTML5-video-shaped surface with the native pipeline underneath: same events, frames in display order with
    the oracle's pixels, a seek restarts at the key-map entry for the time"""
    import hashlib
    from test_pipeline_gpu import oracle_frames
    path = os.path.join(STREAMS, "leon_synth_352x240.jsv")
    want = oracle_frames(open(path, "rb").read())
    out = subprocess.run(["node", "-e", _PIPE_SCRIPT % {"js": JSDIR, "stream": path, "seek": "0.6", "gpu_parser": "true" if gpu_parser else "false"}], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert (r["w"], r["h"]) == (352, 240) and "error" not in r["ev"]
    assert r["ev"][:4] == ["loadstart", "loadedmetadata", "play", "playing"] or r["ev"][:2] == ["loadstart", "loadedmetadata"]
    assert r["ev"].count("ended") == 2 and "seeking" in r["ev"] and "seeked" in r["ev"]
    first, second = r["shown"][:24], r["shown"][24:]
    assert [(s["gop"], s["di"]) for s in first] == [(g, d) for g in range(2) for d in range(12)]
    assert [(s["gop"], s["di"]) for s in second] == [(1, d) for d in range(12)]          # 0.6 s = the second GOP's key entry
    for s in r["shown"]:
        assert s["sha"] == hashlib.sha256(want[(s["gop"], s["di"])].tobytes()).hexdigest(), s
    ts = [s["ts"] for s in first]
    assert ts == sorted(ts)
