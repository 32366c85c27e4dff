"""CPU: the plain-JavaScript oracle (oracle/leon_oracle.js, the CPU baseline of SURVEY.md 8d in the
reference's own language) against the C oracle, bit for bit: planes and RGBA of whole IBBP GOPs,
vectors leaving the picture, a frame crop with odd width, garbage levels over the full int16 range."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ROOT
from oracle import oracle_py as O
import synth as S

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
RUNNER = os.path.join(ROOT, "oracle", "js_baseline.js")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _c_oracle(cw, ch, fw, fh, gop, pics, qm=None):
    outs, res = {}, []
    for ptype, disp, f, b in gop:
        t = pics[disp]
        fwd = f if f is not None else b
        outs[disp] = O.decode_picture(ptype, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                                      repadd=t.get("repadd"), mb_dir=t.get("mb_dir"), mv_fwd=t.get("mv_fwd"),
                                      mv_bwd=t.get("mv_bwd"), qm=qm, ref_fwd=None if fwd is None else outs[fwd],
                                      ref_bwd=None if b is None else outs[b])
        y, cb, cr = O.split_planes(outs[disp], cw, ch)
        res.append({"disp": disp, "type": ptype, "planes": sha(outs[disp]), "rgba": sha(O.ycbcr_to_rgba(y, cb, cr, cw, fw, fh, "cpu"))})
    return res


def _js(tmp_path, cw, ch, fw, fh, gop, pics, qm=None):
    O.dump_gop(str(tmp_path), cw, ch, fw, fh, gop, pics, qm=qm)
    out = subprocess.run(["node", RUNNER, "check", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout)


@pytest.mark.parametrize("cw,ch,fw,fh,in_pic,seed", [(96, 64, 96, 64, True, 1), (176, 144, 171, 141, False, 2), (64, 48, 61, 48, True, 3)])
def test_ibbp_gop_planes_and_rgba(tmp_path, cw, ch, fw, fh, in_pic, seed):
    O.lib()
    rng = np.random.default_rng(seed)
    gop = S.gop_ibbp(6)
    pics = {}
    for ptype, disp, f, b in gop:
        pics[disp] = S.make_picture(rng, cw, ch, ptype, in_picture=in_pic, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
    assert _js(tmp_path, cw, ch, fw, fh, gop, pics) == _c_oracle(cw, ch, fw, fh, gop, pics)


def test_garbage_levels_and_custom_matrices(tmp_path):
    O.lib()
    rng = np.random.default_rng(9)
    cw, ch = 48, 32
    nmb = 6
    qm = rng.integers(1, 256, 128).astype(np.uint8)
    gop = [(1, 0, None, None), (2, 1, 0, None)]
    pics = {}
    for ptype, disp, f, b in gop:
        t = S.make_picture(rng, cw, ch, ptype)
        for k, n in (("coef_y", cw * ch), ("coef_cb", cw * ch // 4), ("coef_cr", cw * ch // 4)):
            t[k] = rng.integers(-32768, 32768, n).astype(np.int16)
        t["qscale"] = rng.integers(0, 32, nmb).astype(np.uint8)
        pics[disp] = t
    assert _js(tmp_path, cw, ch, cw, ch, gop, pics, qm=qm) == _c_oracle(cw, ch, cw, ch, gop, pics, qm=qm)


def test_timing_mode_reports_one_thread_and_workers(tmp_path):
    O.lib()
    rng = np.random.default_rng(4)
    gop = S.gop_ibbp(3)
    pics = {disp: S.make_picture(rng, 64, 48, ptype, force_dir=2 if (ptype == S.PIC_B and f is None) else None) for ptype, disp, f, b in gop}
    O.dump_gop(str(tmp_path), 64, 48, 64, 48, gop, pics)
    out = subprocess.run(["node", RUNNER, "time", str(tmp_path), "0.3", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout)
    assert r["one_thread"]["pictures"] >= 3 and r["workers"]["threads"] == 2 and r["workers"]["pictures"] >= 6
