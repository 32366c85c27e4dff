"""Shared test plumbing: run the same boundary tensors through the oracle (CPU) and
through libleon_hip.so (C ABI, GPU) and compare planes bit for bit."""
import base64
import json
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def b64(s, dtype=np.uint8):
    return np.frombuffer(base64.b64decode(s), dtype=dtype)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def oracle_decode_sequence(O, cw, ch, pics, refs, qm=None):
    """pics: list of tensors dicts in coded order, each with keys 'slot', 'ref_fwd', 'ref_bwd'
    (indices into the `out` dict).  Returns {slot: flat [Y|Cb|Cr] uint8}."""
    out = dict(refs)
    for t in pics:
        out[t["slot"]] = O.decode_picture(
            t["type"], cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
            repadd=t.get("repadd"), mb_dir=t.get("mb_dir"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
            qm=qm, ref_fwd=None if t.get("ref_fwd") is None else out[t["ref_fwd"]],
            ref_bwd=None if t.get("ref_bwd") is None else out[t["ref_bwd"]])
    return out


def hip_submit(L, dec, t, keep, rgba_out=None):
    """rgba_out (a device pointer): the fused path, k_recon_display -- planes AND the RGBA frame from one launch"""
    extra = {} if rgba_out is None else dict(rgba_out=rgba_out, no_planes=False)
    p = L.make_picture(t["type"], t["slot"], t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"],
                       repadd=t.get("repadd"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                       mb_dir=t.get("mb_dir"),
                       ref_fwd_slot=-1 if t.get("ref_fwd") is None else t["ref_fwd"],
                       ref_bwd_slot=-1 if t.get("ref_bwd") is None else t["ref_bwd"], keep=keep, **extra)
    dec.submit_picture(p)


def planes_flat(y, cb, cr):
    return np.concatenate([y.ravel(), cb.ravel(), cr.ravel()])


def hip_submit_sparse(L, dec, t, keep, cw, ch, rgba_out=None):
    """The same picture through the sparse boundary (include/leon_vlc.h lists)."""
    extra = {} if rgba_out is None else dict(rgba_out=rgba_out, no_planes=False)
    import leon_vlc_ctypes as V
    if "grp_off" in t and t.get("entries") is not None:
        grp_off, entries = t["grp_off"], t["entries"]
    else:
        grp_off, entries = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], cw, ch)
    p = L.make_sparse_picture(t["type"], t["slot"], grp_off, entries, len(entries), t["qscale"], t["intra"],
                              repadd=t.get("repadd"), mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                              mb_dir=t.get("mb_dir"),
                              ref_fwd_slot=-1 if t.get("ref_fwd") is None else t["ref_fwd"],
                              ref_bwd_slot=-1 if t.get("ref_bwd") is None else t["ref_bwd"], keep=keep, **extra)
    dec.submit_sparse([p], L.MEM_HOST)
