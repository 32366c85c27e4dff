#!/usr/bin/env python3
"""make_glsl_cases.py -- INPUTS of the tensor-driven golden cases (seeded, deterministic).

Writes the boundary tensors that tools/make_golden_glsl.js then drives through the
reference's own jsv.prototype.IDCT_GL (decoders/jsv.js:1177-1336) on tools/softgl.  This
script produces inputs only; the expected outputs come from executing the reference.

    python3 tools/make_glsl_cases.py /tmp/glsl_cases.json
"""
import base64
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")]
import synth as S  # noqa: E402

# the custom intra matrix of tests/golden/streams/custom_intra_ip_48x32.jsv (tools/make_streams.py):
# small entries (a dequantised value can floor to 0 -> the shader's 0 -> +1 step) and 255s (clamps)
CUSTOM_INTRA = np.array([8] + [1 + (k * 37) % 7 for k in range(1, 32)] + [255 - (k % 5) for k in range(32, 64)], dtype=np.uint8)


def z(a):
    return base64.b64encode(zlib.compress(np.ascontiguousarray(a).tobytes(), 9)).decode()


def pack(t):
    out = {"type": int(t["type"])}
    for k in ("coef_y", "coef_cb", "coef_cr"):
        out[k] = z(np.asarray(t[k], dtype="<i2"))
    for k in ("qscale", "intra", "repadd"):
        if k in t:
            out[k] = z(np.asarray(t[k], dtype=np.uint8))
    if "mv_fwd" in t:
        out["mv_fwd"] = z(np.asarray(t["mv_fwd"], dtype="<i2"))
    return out


def garbage(rng, cw, ch, ptype, flags=(0, 1, 127, 128, 255)):
    nmb = (cw // 16) * (ch // 16)
    t = {"type": ptype,
         "coef_y": rng.integers(-32768, 32768, size=(ch, cw)).astype(np.int16),
         "coef_cb": rng.integers(-32768, 32768, size=(ch // 2, cw // 2)).astype(np.int16),
         "coef_cr": rng.integers(-32768, 32768, size=(ch // 2, cw // 2)).astype(np.int16),
         "qscale": rng.integers(0, 32, size=nmb).astype(np.uint8),
         "intra": rng.choice(np.array(flags, dtype=np.uint8), size=nmb)}
    # thin the planes out in places: whole zero blocks and zero rows exercise the `continue` path
    for p in (t["coef_y"], t["coef_cb"], t["coef_cr"]):
        m = rng.random((p.shape[0] // 8, p.shape[1] // 8)) < 0.3
        p[np.kron(m, np.ones((8, 8), dtype=bool))] = 0
        p[rng.random(p.shape) < 0.2] = 0
    if ptype == S.PIC_P:
        t["repadd"] = rng.choice(np.array(flags, dtype=np.uint8), size=nmb)
        t["mv_fwd"] = rng.integers(-70, 71, size=nmb * 2).astype(np.int16)      # leaves the picture on purpose
    return t


def single_coefs(cw, ch, ptype, rng, level_set=(1, -1, 2, -2, 3, -3, 255, -255)):
    """one non-zero coefficient per block, position = block index mod 64: every basis function alone"""
    nmb = (cw // 16) * (ch // 16)
    t = {"type": ptype, "qscale": ((np.arange(nmb) * 7) % 31 + 1).astype(np.uint8),
         "intra": np.full(nmb, 255 if ptype == S.PIC_I else 0, dtype=np.uint8)}
    n = 0
    for k, (W, H) in zip(("coef_y", "coef_cb", "coef_cr"), ((cw, ch), (cw // 2, ch // 2), (cw // 2, ch // 2))):
        p = np.zeros((H, W), dtype=np.int16)
        for by in range(H // 8):
            for bx in range(W // 8):
                pos = n % 64
                p[8 * by + pos // 8, 8 * bx + pos % 8] = level_set[(n // 64 + n) % len(level_set)]
                if ptype == S.PIC_I and pos != 0:
                    p[8 * by, 8 * bx] = (n * 13) % 256          # intra DC in the predictor domain
                n += 1
        t[k] = p
    if ptype == S.PIC_P:
        t["repadd"] = np.zeros(nmb, dtype=np.uint8)
        t["mv_fwd"] = np.zeros(nmb * 2, dtype=np.int16)
    return t


def dc_only(cw, ch, values):
    nmb = (cw // 16) * (ch // 16)
    t = {"type": S.PIC_I, "qscale": np.full(nmb, 8, dtype=np.uint8), "intra": np.full(nmb, 255, dtype=np.uint8)}
    n = 0
    for k, (W, H) in zip(("coef_y", "coef_cb", "coef_cr"), ((cw, ch), (cw // 2, ch // 2), (cw // 2, ch // 2))):
        p = np.zeros((H, W), dtype=np.int16)
        for by in range(H // 8):
            for bx in range(W // 8):
                p[8 * by, 8 * bx] = values[n % len(values)]
                n += 1
        t[k] = p
    return t


def clamp_case(cw, ch, ptype, rng):
    """levels at the escape-code extremes with the extreme quantiser scales: +-2047 / -2048 clamps,
    and (with the custom matrix's entries 1..7) products that floor to 0"""
    nmb = (cw // 16) * (ch // 16)
    t = {"type": ptype, "qscale": rng.choice(np.array([1, 2, 31], dtype=np.uint8), size=nmb),
         "intra": np.full(nmb, 255, dtype=np.uint8) if ptype == S.PIC_I else rng.choice(np.array([0, 255], dtype=np.uint8), size=nmb)}
    for k, (W, H) in zip(("coef_y", "coef_cb", "coef_cr"), ((cw, ch), (cw // 2, ch // 2), (cw // 2, ch // 2))):
        p = rng.choice(np.array([0, 0, 0, 1, -1, 2, -2, 3, 255, -255, 127, -128], dtype=np.int16), size=(H, W))
        p[::8, ::8] = rng.integers(0, 256, size=(H // 8, W // 8))
        t[k] = p
    if ptype == S.PIC_P:
        t["repadd"] = t["intra"].copy()
        t["mv_fwd"] = S.clip_vectors(rng.integers(-31, 32, size=nmb * 2).astype(np.int16), cw // 16, ch // 16, cw, ch)
    return t


def main(out):
    cases = []
    rng = np.random.default_rng(0x474C534C)
    cw, ch = 96, 64
    seq = [S.make_picture(rng, cw, ch, S.PIC_I)] + [S.make_picture(rng, cw, ch, S.PIC_P) for _ in range(3)]
    cases.append({"name": "synthetic_ippp_default_matrices", "stream": "slices5_ip_96x64.jsv", "pictures": [pack(t) for t in seq]})
    seq = [garbage(rng, cw, ch, S.PIC_I), garbage(rng, cw, ch, S.PIC_P), garbage(rng, cw, ch, S.PIC_P), garbage(rng, cw, ch, S.PIC_I)]
    cases.append({"name": "full_int16_garbage_vectors_leave_picture", "stream": "slices5_ip_96x64.jsv", "pictures": [pack(t) for t in seq]})
    seq = [single_coefs(cw, ch, S.PIC_I, rng), single_coefs(cw, ch, S.PIC_P, rng), dc_only(cw, ch, [0, 255, 128, 1, 254, 16, 235])]
    cases.append({"name": "single_coefficients_and_dc", "stream": "slices5_ip_96x64.jsv", "pictures": [pack(t) for t in seq]})
    cw, ch = 48, 32
    seq = [clamp_case(cw, ch, S.PIC_I, rng), clamp_case(cw, ch, S.PIC_P, rng), single_coefs(cw, ch, S.PIC_I, rng),
           garbage(rng, cw, ch, S.PIC_P), S.make_picture(rng, cw, ch, S.PIC_P, qm_intra=CUSTOM_INTRA.reshape(8, 8))]
    cases.append({"name": "custom_intra_matrix_clamps_zero_to_one", "stream": "custom_intra_ip_48x32.jsv", "pictures": [pack(t) for t in seq]})
    with open(out, "w") as f:
        json.dump(cases, f)
    print(len(cases), "cases ->", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/glsl_cases.json")
