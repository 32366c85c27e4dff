"""yuva streams (container flag `a`, decoders/jsv.js:256-259): the repo's macroblock syntax for the fourth
component (tools/jsv_writer.py write_picture) read back by both product parsers -- the JavaScript one and
the native front end -- equals what the writer encoded; on the GPU, through Node, the decoded planes and
RGBA (A from the plane) equal the oracle's.  The reference defines no syntax and decodes no fourth
component, so these are round trips and oracle comparisons, not reference pins."""
import hashlib
import os
import shutil

import numpy as np
import pytest

from helpers import ROOT

STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
NAME = "yuva_ibbp_96x64"
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def source_pictures():
    """the tensors tools/make_streams.py fed to the writer for yuva_ibbp_96x64.jsv"""
    import synth as S
    rng = np.random.default_rng(404)
    out = []
    for gop in (S.gop_ibbp(9), S.gop_ippp(4)):
        for ptype, disp, f, b in gop:
            t = S.make_picture(rng, 96, 64, ptype, alpha=True, force_dir=2 if (ptype == S.PIC_B and f is None) else None)
            t["gop_entry"] = (ptype, disp, f, b)
            out.append(t)
    return out


def test_native_front_end_reads_the_alpha_blocks():
    import leon_vlc_ctypes as V
    data = open(os.path.join(STREAMS, NAME + ".jsv"), "rb").read()
    src = source_pictures()
    for threads in (1, 4):
        st = V.Stream(data, threads=threads)
        assert st.info.has_alpha == 1 and st.info.n_groups == 2 * (2 * 4 * 2) + 2 * (4 * 1)
        n = 0
        while True:
            p = st.next_picture(dense=True)
            if p is None:
                break
            t = src[n]
            for k in ("coef_y", "coef_cb", "coef_cr", "coef_a"):
                assert np.array_equal(p[k].reshape(-1), t[k].reshape(-1)), (n, k)
            go, en = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], 96, 64, coef_a=t["coef_a"])
            assert np.array_equal(go, p["grp_off"]) and np.array_equal(np.sort(en), np.sort(p["entries"])), n
            n += 1
        assert n == len(src)


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_javascript_parser_reads_the_alpha_blocks():
    from test_js_parser import run_cli
    src = source_pictures()
    for extra in ((), ("--native",)):
        got = run_cli("tensors", os.path.join(STREAMS, NAME + ".jsv"), *extra)
        assert [e for e in got["events"] if e["ev"] == "meta"][0]["a"] == 1
        assert len(got["pictures"]) == len(src)
        for i, (p, t) in enumerate(zip(got["pictures"], src)):
            assert p["type"] == t["type"], i
            assert p["sha"]["coefA"] == sha(t["coef_a"].astype("<i2")), "A coefficients of picture %d %s" % (i, extra)
            assert p["sha"]["coefY"] == sha(t["coef_y"].astype("<i2")) and p["sha"]["coefCr"] == sha(t["coef_cr"].astype("<i2")), i


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
@pytest.mark.parametrize("extra", [(), ("--native",)], ids=["js-parser-dense", "native-sparse"])
def test_yuva_stream_through_node_equals_oracle(extra):
    from oracle import oracle_py as O
    from test_js_parser import run_cli
    src = source_pictures()
    cw, ch = 96, 64
    outs, old, new = [], None, None
    for t in src:
        ptype = t["type"]
        if ptype == 1:
            old = new = None
        fwd = new if ptype == 2 else (old if old is not None else new)
        out = O.decode_picture(ptype, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], repadd=t.get("repadd"),
                               mb_dir=t.get("mb_dir"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"), ref_fwd=fwd, ref_bwd=new,
                               coef_a=t["coef_a"])
        if ptype != 3:
            old, new = new, out
        outs.append(out)
    got = run_cli("decode", os.path.join(STREAMS, NAME + ".jsv"), "--rgba", *extra)
    assert got["events"][-1]["ev"] == "ended" and len(got["pictures"]) == len(src)
    n = cw * ch
    n3 = n * 3 // 2
    for i, (p, e) in enumerate(zip(got["pictures"], outs)):
        assert p["planes"]["y"] == sha(e[:n]) and p["planes"]["cb"] == sha(e[n:n + n // 4]) and p["planes"]["cr"] == sha(e[n + n // 4:n3]), i
        assert p["planes"]["a"] == sha(e[n3:]), "A plane of picture %d" % i
        y, cb, cr = O.split_planes(e[:n3], cw, ch)
        assert p["rgba"] == sha(O.ycbcr_to_rgba(y, cb, cr, cw, cw, ch, "cpu", a=e[n3:])), "RGBA of picture %d" % i
