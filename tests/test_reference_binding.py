"""CPU, build container only: the shipped binding (mpeg1video-decoder-webgl_amd/js/reference_binding.js) applied to the
UNMODIFIED reference decoder object -- /root/reference/decoders/jsv.js loaded in a vm the way tools/make_golden.js loads
it -- with a recording stand-in for the N-API addon (tests/reference_binding_check.js).  The reference's own decodeFrame
loop decodes the fixture streams; what the binding hands to `submitPicture` must be, field for field, what the same
reference uploaded through WebGL (tests/golden/parser_*.json: jsv.prototype.IDCT_GL's texImage2D calls,
decoders/jsv.js:1204-1298, recorded through a fake gl).  Only this test and the binding travel; /root/reference does
not, so the test skips where it is absent (the GPU box).

The addon behind the stand-in's surface is exercised on the GPU by tests/test_js_e2e_gpu.py (through this repository's
own decoder object, which makes the same calls).
"""
import json
import os
import shutil
import subprocess

import pytest

from helpers import ROOT, load_golden

REF = "/root/reference"
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
pytestmark = [pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed"),
              pytest.mark.skipif(not os.path.exists(os.path.join(REF, "decoders", "jsv.js")), reason="the reference is not on this machine")]


def run_check(stream, *flags):
    out = subprocess.run(["node", os.path.join(ROOT, "tests", "reference_binding_check.js"), REF, os.path.join(STREAMS, stream + ".jsv")] + list(flags),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout)


def uploads_of(pic):
    """the reference's uploads of one picture, by the field of the boundary they are"""
    ups = [u for u in pic["uploads"] if "unit" in u]
    unit = lambda n: [u for u in ups if u["unit"] == n]
    d = {"coefY": unit(0)[0], "coefCb": unit(0)[1], "coefCr": unit(0)[2], "qscale": unit(2)[0], "intra": unit(4)[0]}
    if pic["type"] == 2:
        d["repadd"], d["mvFwd"] = unit(2)[1], unit(3)[0]
    return d


@pytest.mark.parametrize("name", ["tiny_ip_32x32", "slices5_ip_96x64", "leon_synth_352x240", "ibbp_96x64"])
def test_submit_picture_gets_what_the_reference_uploads(name):
    got = run_check(name, "--seek")
    ref = load_golden("parser_%s.json" % name)
    assert (got["mbWidth"], got["mbHeight"], got["codedWidth"], got["codedHeight"]) == (ref["mbWidth"], ref["mbHeight"], ref["codedWidth"], ref["codedHeight"])
    # same events in the same order, same time stamps: the binding does not disturb the decode loop
    assert [(e["ev"], e.get("ts")) for e in got["events"]] == [(e["ev"], e.get("ts")) for e in ref["events"]]
    calls = got["calls"]
    create = [c for c in calls if c["call"] == "create"]
    assert len(create) == 1 and create[0]["cfg"]["codedWidth"] == ref["codedWidth"] and create[0]["cfg"]["nSlots"] == got["nSlots"] == 13
    subs = [c for c in calls if c["call"] == "submitPicture"]
    assert len(subs) == len(ref["pictures"]) > 0
    mbs = ref["mbWidth"] * ref["mbHeight"]
    for i, (s, r) in enumerate(zip(subs, ref["pictures"])):
        assert s["type"] == r["type"], i
        want = uploads_of(r)
        fields = sorted(want) + ["outSlot", "refFwdSlot", "type"]
        assert s["keys"] == sorted(fields), "picture %d hands over %s" % (i, s["keys"])
        for k, u in want.items():
            assert s[k]["sha256"] == u["sha256"], "picture %d: %s differs from the reference's upload" % (i, k)
            elem = 2 if k.startswith("coef") or k == "mvFwd" else 1
            assert s[k]["length"] * elem == u["bytes"], (i, k)
            assert s[k]["ctor"] == ("Int16Array" if elem == 2 else "Uint8Array"), (i, k)
        assert s["qscale"]["length"] == mbs
        # slots: never the picture's own reference; P pictures predict from the picture submitted before them (jsv.js:665)
        assert s["outSlot"] != s["refFwdSlot"]
        assert s["refFwdSlot"] == (subs[i - 1]["outSlot"] if s["type"] == 2 else -1), i
    # every frame event carries the slot its picture was written to, one shared texture object for the three components
    assert [f["slot"] for f in got["frames"]] == [s["outSlot"] for s in subs]
    assert all(f["sameTexture"] for f in got["frames"])
    # slot bookkeeping: released exactly once each, the forward reference only after the picture that reads it was
    # submitted, and at most two slots busy at any time with a display that releases at once
    busy, peak = set(), 0
    for c in calls:
        if c["call"] == "acquireSlot":
            assert c["slot"] not in busy
            busy.add(c["slot"])
            peak = max(peak, len(busy))
        elif c["call"] == "releaseSlot":
            assert c["slot"] in busy
            busy.remove(c["slot"])
        elif c["call"] == "submitPicture" and c["type"] == 2:
            assert c["refFwdSlot"] in busy, "the reference slot was given back before its reader was submitted"
        elif c["call"] == "freeDecodedSlots":
            busy.clear()
    assert peak <= 2 and calls[-1]["call"] == "freeDecodedSlots"


def test_default_matrices_reach_the_library_once():
    got = run_check("tiny_ip_32x32")
    qm = [c for c in got["calls"] if c["call"] == "setQuantMatrices"]
    import hashlib
    import synth as S
    assert len(qm) == 1
    assert qm[0]["intra"] == hashlib.sha256(S.DEFAULT_INTRA_QUANT.tobytes()).hexdigest()
    assert qm[0]["nonIntra"] == hashlib.sha256(S.DEFAULT_NON_INTRA_QUANT.tobytes()).hexdigest()


def test_custom_matrices(tmp_path):
    """a sequence header with custom matrices: the intra one reaches the library; the non-intra one only when asked for
    (the reference's own GL path drops it, decoders/jsv.js:556 binds a texture that was never created)"""
    import hashlib
    import numpy as np
    import jsv_writer as JW
    import synth as S
    rng = np.random.default_rng(7)
    qi = rng.integers(1, 256, 64).astype(np.uint8)
    qi[0] = 8
    qn = rng.integers(1, 256, 64).astype(np.uint8)
    cw = ch = 32
    pics = [S.make_picture(rng, cw, ch, S.PIC_I, qm_intra=qi, qm_non=qn), S.make_picture(rng, cw, ch, S.PIC_P, qm_intra=qi, qm_non=qn)]
    pics[0]["display"], pics[1]["display"] = 0, 1
    data, _ = JW.write_stream(pics, cw, ch, qm_intra=qi, qm_non_intra=qn)
    p = tmp_path / "custom.jsv"
    p.write_bytes(data)
    sha = lambda a: hashlib.sha256(np.asarray(a, np.uint8).tobytes()).hexdigest()
    for flags, want_non in (((), S.DEFAULT_NON_INTRA_QUANT), (("--honour-non-intra",), qn)):
        out = subprocess.run(["node", os.path.join(ROOT, "tests", "reference_binding_check.js"), REF, str(p)] + list(flags),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        got = json.loads(out.stdout)
        qm = [c for c in got["calls"] if c["call"] == "setQuantMatrices"]
        assert qm[-1]["intra"] == sha(qi) and qm[-1]["nonIntra"] == sha(want_non), flags
        assert len([c for c in got["calls"] if c["call"] == "submitPicture"]) == 2
