"""CPU: conformance tests of the GLSL ES 1.00 evaluator and the software GL machine (tools/softgl/) that pin the
oracle's dequantisation + IDCT by executing the reference's own shader text (tools/make_golden_glsl.js; the shaders:
/root/reference decoders/shaders/mpeg1video.js:18-29, player/parts/end.js:77-166).

The pin rests on that interpreter, so the interpreter gets tests of its own: small programs whose results are
derived by hand here (or with numpy's binary32 arithmetic, an implementation that shares nothing with the evaluator),
none of them taken from the reference's shaders.

Three things are NOT facts of GLSL ES 1.00 but decisions of the arithmetic model, fixed by fiat and
asserted below so that a change of the model cannot go unnoticed:
  D1  `int` is 32-bit two's complement and `/` truncates toward zero (the language leaves both open);
  D2  `float` is IEEE-754 binary32, every operation rounded to nearest-even on its own, no contraction;
  D5  a colour written to an RGBA8 attachment is clamp(c, 0, 1) * 255 rounded to nearest, ties to even.
Also by fiat: `vec4 * mat4` sums its four products left to right (the reference's display shader multiplies that way,
player/parts/end.js:87-92; the order is the compiler's in a real driver -- D10 reports the canvas, it does not gate it),
and the sampler resolves a coordinate to 8 fractional bits before the floor (D8).
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ROOT

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")

f32 = np.float32


def frag(body, out, decl=""):
    return {"kind": "glsl", "src": "precision highp float;\n%s\nvoid main() {\n%s\ngl_FragColor = %s;\n}\n" % (decl, body, out)}


def run(cases, tmp_path):
    p = tmp_path / "cases.json"
    p.write_text(json.dumps(cases))
    out = subprocess.run(["node", os.path.join(ROOT, "tests", "softgl_runner.js"), str(p)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout)
    assert len(res) == len(cases)
    return res


def fl(x):
    return float(f32(x))


# ---- (name, case, expected gl_FragColor) ---------------------------------------------------------------------------
GLSL_VALUE_CASES = [
    # D1: int division truncates toward zero, all four sign combinations; int() truncates
    ("int div -7/2", frag("int a = -7; int b = 2;", "vec4(float(a / b), float(7 / -2), float(-7 / -2), float(7 / 2))"), [-3, -3, 3, 3]),
    ("int div small", frag("int a = -1; int b = 256;", "vec4(float(a / b), float(-255 / 256), float(-256 / 256), float(-257 / 256))"), [0, 0, -1, -1]),
    ("int() truncates", frag("", "vec4(float(int(-2.7)), float(int(2.7)), float(int(-0.5)), float(int(-3.0)))"), [-2, 2, 0, -3]),
    ("int wraps mod 2^32", frag("int a = 2147483647; int b = a + 1; int c = 65536 * 65536; int d = -a - 2;", "vec4(float(b), float(c), float(d), 0.0)"),
     [-2147483648.0, 0, 2147483648.0, 0]),      # d = 2^31 - 1 after the wrap, which binary32 rounds to 2^31
    ("int compound assign", frag("int a = 7; a /= 2; int b = -7; b /= 2; int c = 5; c *= -3; int d = 1; d -= 4;", "vec4(float(a), float(b), float(c), float(d))"), [3, -3, -15, -3]),
    ("int ++ --", frag("int i = 3; int a = i++; int b = ++i; int c = i--; int d = --i;", "vec4(float(a), float(b), float(c), float(d))"), [3, 5, 5, 3]),
    # mod as 8.3 defines it: x - y * floor(x / y), also for negative x and negative y
    ("mod", frag("", "vec4(mod(-1.0, 4.0), mod(5.5, 2.0), mod(-5.5, 2.0), mod(7.0, -3.0))"), [3.0, 1.5, 0.5, -2.0]),
    ("mod vec", frag("vec2 m = mod(vec2(9.0, -9.0), 4.0); vec2 n = mod(vec2(9.0, -9.0), vec2(5.0, 2.0));", "vec4(m, n)"), [1.0, 3.0, 4.0, 1.0]),
    ("floor sign", frag("", "vec4(floor(-0.5), floor(2.0), sign(-3.5), sign(0.0))"), [-1, 2, -1, 0]),
    ("abs min max", frag("", "vec4(abs(-2.5), min(1.0, -2.0), max(1.0, -2.0), fract(-0.25))"), [2.5, -2, 1, 0.75]),
    ("clamp", frag("", "vec4(clamp(5.0, 0.0, 1.0), clamp(-5.0, 0.0, 1.0), clamp(0.25, 0.0, 1.0), floor(-0.0))"), [1, 0, 0.25, 0]),
    ("vector built-ins", frag("vec4 v = floor(vec4(-1.5, 1.5, -0.0, 2.999)); vec4 w = max(v, 0.0);", "v + w * 10.0"), [-2, 11, 0, 22]),
    # D2: binary32, every operation rounded on its own
    ("2^24 + 1 + 1", frag("float a = 16777216.0; float b = (a + 1.0) + 1.0; float c = a + (1.0 + 1.0);", "vec4(b, c, a + 1.0, a + 3.0)"),
     [16777216.0, 16777218.0, 16777216.0, 16777220.0]),
    ("0.1 + 0.2", frag("", "vec4(0.1 + 0.2, 0.1 * 3.0, 1.0 / 3.0, 0.4)"),
     [float(f32(0.1) + f32(0.2)), float(f32(0.1) * f32(3.0)), float(f32(1.0) / f32(3.0)), fl(0.4)]),
    ("no contraction", frag("float a = 1.000244140625; float c = -1.00048828125; float p = a * a;", "vec4(p + c, p, a * a + c, 0.0)"),
     # a = 1 + 2^-12: a*a = 1 + 2^-11 + 2^-24, which rounds to 1 + 2^-11 (a tie, to even); a fused multiply-add would leave 2^-24
     [0.0, 1.00048828125, 0.0, 0.0]),
    ("literal rounding", frag("float y = 0.4; float z = 16777217.0;", "vec4(y * 5.0, floor(15.0 * y), z, float(16777217))"),
     [float(f32(0.4) * f32(5.0)), 6.0, 16777216.0, 16777216.0]),
    ("division by 0.4", frag("float y = 0.4;", "vec4(float(int(3.0 / y)), float(int(-3.0 / y)), float(int(7.0 / y)), float(int(-1.0 / y)))"), [7, -7, 17, -2]),
    ("sqrt sin cos rounded", frag("", "vec4(sqrt(2.0), cos(0.0), sin(0.0), sqrt(16.0))"), [fl(np.sqrt(2.0)), 1.0, 0.0, 4.0]),
    ("dot left to right", frag("vec4 a = vec4(100000000.0, 1.0, -100000000.0, 1.0); vec4 o = vec4(1.0, 1.0, 1.0, 1.0);", "vec4(dot(a, o), dot(a.xzyw, o), dot(vec2(3.0, 4.0), vec2(3.0, 4.0)), 0.0)"),
     [1.0, 2.0, 25.0, 0.0]),
    # vec4 * mat4: component j = dot(v, column j); the constructor fills columns first
    ("vec4 * mat4 columns", frag("mat4 m = mat4(1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 16.0); vec4 v = vec4(1.0, 2.0, 3.0, 4.0);", "v * m"),
     [30.0, 70.0, 110.0, 150.0]),
    ("vec4 * mat4 order", frag("mat4 m = mat4(1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0);", "vec4(100000000.0, 1.0, -100000000.0, 1.0) * m"),
     [1.0, 1.0, 0.0, 0.0]),
    # swizzles as r- and l-values; containers are values
    ("swizzle write", frag("vec4 v = vec4(1.0, 2.0, 3.0, 4.0); v.xz = vec2(9.0, 8.0);", "v"), [9, 2, 8, 4]),
    ("swizzle read", frag("vec4 v = vec4(1.0, 2.0, 3.0, 4.0);", "v.wzyx + v.rrgg * 10.0"), [14, 13, 22, 21]),
    ("swizzle swap", frag("vec4 v = vec4(1.0, 2.0, 3.0, 4.0); v.yx = v.xy; v.a = v.r;", "v"), [2, 1, 3, 2]),
    ("swizzle stpq and index", frag("vec4 v = vec4(1.0, 2.0, 3.0, 4.0); v[2] = v.q; int i = 1; v[i] += 0.5;", "vec4(v.s, v.t, v.p, v[3])"), [1, 2.5, 4, 4]),
    ("assignment copies", frag("vec2 a = vec2(1.0, 2.0); vec2 b = a; b.x = 5.0; float arr[3]; arr[0] = 7.0; float c = arr[0]; arr[0] = 8.0;", "vec4(a.x, b.x, c, arr[0])"), [1, 5, 7, 8]),
    ("array of float", frag("float a[4]; a[0] = 1.0; a[3] = 4.0; int i = 3; a[1] = a[i] * 2.0; a[i - 1] = a[0];", "vec4(a[0], a[1], a[2], a[3])"), [1, 8, 1, 4]),
    ("arrays start at zero", frag("float a[3]; int b; bool c; vec2 d;", "vec4(a[2], float(b), float(c), d.y)"), [0, 0, 0, 0]),
    # control flow
    ("for continue break", frag("float s = 0.0; for (int i = 0; i < 10; i++) { if (i == 2) continue; if (i == 5) break; s += float(i); }", "vec4(s, 0.0, 0.0, 0.0)"), [8, 0, 0, 0]),
    ("nested loops", frag("int n = 0; for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) { if (j > i) break; n += 1; }", "vec4(float(n), 0.0, 0.0, 0.0)"), [6, 0, 0, 0]),
    ("ternary and logic", frag("float a = 3.0 > 2.0 ? 1.0 : 2.0; bool b = (1 < 2) && !(2 < 1) || false; float c = b ? 5.0 : 6.0; bool e = vec2(1.0, 2.0) == vec2(1.0, 2.0);", "vec4(a, c, float(e), float(1 != 1))"), [1, 5, 1, 0]),
    ("functions take values", frag("float x = 1.0; float y = bump(x); vec2 p = vec2(1.0, 2.0); float z = first(p);", "vec4(x, y, z, p.x)",
                                   decl="float bump(float v) { v += 1.0; return v; }\nfloat first(vec2 q) { q.x = 9.0; return q.x; }"), [1, 2, 9, 1]),
    ("early return", frag("", "vec4(pick(1), pick(-1), pick(0), 0.0)", decl="float pick(int i) { if (i > 0) return 1.0; if (i < 0) { return -1.0; } return 0.5; }"), [1, -1, 0.5, 0]),
    ("globals const uniform", frag("g = g + k;", "vec4(g, k, u, float(n))", decl="const float k = 2.5;\nfloat g = 1.0;\nuniform float u;\nuniform int n;"), [3.5, 2.5, 0, 0]),
    ("constructors", frag("vec4 a = vec4(vec2(1.0, 2.0), 3.0, 4.0); vec3 b = vec3(7.0); vec2 c = vec2(a); float d = float(true);", "vec4(a.z + b.y, c.y, d, float(int(bool(2))))"), [10, 2, 1, 1]),
    ("int vector constructor rounds", frag("vec2 a = vec2(16777217, 3);", "vec4(a, 0.0, 0.0)"), [16777216.0, 3, 0, 0]),
    ("comments and precision", frag("/* block */ float a = 1.0; // line\n mediump float b = 2.0;", "vec4(a, b, 0.0, 0.0)", decl="precision mediump int;"), [1, 2, 0, 0]),
    ("unary minus and grouping", frag("int a = -(3 - 5) * 2; float b = -(1.5 - 2.0) * 2.0; float c = 2.0 + 3.0 * 4.0 - 6.0 / 3.0;", "vec4(float(a), b, c, float(7 - 2 - 1))"), [4, 1, 12, 4]),
]

# ---- programs a GLSL ES 1.00 compiler must reject (no implicit conversions, 4.1.10 / 5.9), and run-time faults -------
GLSL_ERROR_CASES = [
    ("float from int literal", frag("float x = 1;", "vec4(x)"), "initialiser"),
    ("int from float literal", frag("int i = 1.0;", "vec4(0.0)"), "initialiser"),
    ("int + float", frag("float x = 1 + 1.0;", "vec4(x)"), "operator"),
    ("float * int", frag("float x = 2.0 * 3;", "vec4(x)"), "operator"),
    ("vec * int", frag("vec2 v = vec2(1.0, 2.0) * 2;", "vec4(v, v)"), "operator"),
    ("if on int", frag("if (1) { }", "vec4(0.0)"), "condition"),
    ("compare int with float", frag("bool b = 1 < 2.0;", "vec4(0.0)"), "<"),
    ("assign float to int", frag("int i = 0; i = 2.0;", "vec4(0.0)"), "assign"),
    ("float index", frag("float a[2]; a[1.0] = 0.0;", "vec4(0.0)"), "index"),
    ("duplicate swizzle l-value", frag("vec4 v = vec4(0.0); v.xx = vec2(1.0, 2.0);", "v"), "l-value"),
    ("swizzle beyond vec2", frag("vec2 v = vec2(0.0);", "vec4(v.z)"), "swizzle"),
    ("mixed swizzle sets", frag("vec4 v = vec4(0.0);", "vec4(v.xg)"), "swizzle"),
    ("undeclared", frag("", "vec4(nope)"), "undeclared"),
    ("wrong argument type", frag("", "vec4(floor(1))"), "overload"),
    ("return type", frag("", "vec4(f())", decl="float f() { return 1; }"), "return"),
    ("write to a uniform", frag("u = 1.0;", "vec4(u)", decl="uniform float u;"), "l-value"),
    ("array index out of range", frag("float a[2]; int i = 2; a[i] = 1.0;", "vec4(0.0)"), "out of range"),
    ("integer division by zero", frag("int z = 0; int a = 1 / z;", "vec4(0.0)"), "division by zero"),
]


def test_values(tmp_path):
    res = run([c for _, c, _ in GLSL_VALUE_CASES], tmp_path)
    bad = []
    for (name, _, want), got in zip(GLSL_VALUE_CASES, res):
        if "error" in got or [float(x) for x in got["color"]] != [float(w) for w in want]:
            bad.append((name, got, want))
    assert not bad, bad
    assert len(GLSL_VALUE_CASES) + len(GLSL_ERROR_CASES) >= 40


def test_rejected_programs(tmp_path):
    res = run([c for _, c, _ in GLSL_ERROR_CASES], tmp_path)
    bad = [(name, got) for (name, _, needle), got in zip(GLSL_ERROR_CASES, res) if "error" not in got or needle not in got["error"]]
    assert not bad, bad


def test_the_fused_form_would_differ():
    """the 'no contraction' case above is a real discriminator: a fused multiply-add gives another answer"""
    a, c = np.float64(1.000244140625), np.float64(-1.00048828125)
    assert float(f32(f32(f32(a) * f32(a)) + f32(c))) == 0.0
    assert float(f32(a * a + c)) == 2.0 ** -24


# ---- the GL machine: sampling, texture formats, varyings, the RGBA8 store ------------------------------------------

def gl_case(fs, w, h, textures=(), **kw):
    return dict(kind="gl", fs="precision highp float;\nvarying vec2 uv;\n" + fs, w=w, h=h, textures=list(textures), **kw)


def test_sampling_and_store(tmp_path):
    # a 4 x 1 RGBA texture whose texels are told apart by their red byte
    tex4 = dict(unit=0, w=4, h=1, fmt="rgba", bytes=[10, 0, 0, 255, 20, 0, 0, 255, 30, 0, 0, 255, 40, 0, 0, 255])
    sample = lambda u: gl_case("uniform sampler2D t0; uniform float u; void main() { gl_FragColor = texture2D(t0, vec2(u, 0.5)); }", 1, 1, [tex4], uniforms1f={"u": u})
    cases = [
        sample(0.0), sample(0.125), sample(0.249), sample(0.25), sample(0.2499999), sample(0.2501), sample(0.5), sample(0.74),
        sample(0.999), sample(1.0), sample(1.7), sample(-0.3),
        # NEAREST over a 2 x 2 texture from the interpolated varying: every fragment reads its own texel; row 0 is the first row uploaded
        gl_case("uniform sampler2D t0; void main() { gl_FragColor = texture2D(t0, uv); }", 2, 2,
                [dict(unit=0, w=2, h=2, fmt="rgba", bytes=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])]),
        # LUMINANCE -> (L, L, L, 1); LUMINANCE_ALPHA -> (L, L, L, A); rows padded to UNPACK_ALIGNMENT 4 / packed at 1
        gl_case("uniform sampler2D t0; void main() { gl_FragColor = texture2D(t0, uv); }", 3, 2,
                [dict(unit=0, w=3, h=2, fmt="l", bytes=[1, 2, 3, 99, 4, 5, 6, 99])]),
        gl_case("uniform sampler2D t0; void main() { gl_FragColor = texture2D(t0, uv); }", 3, 2,
                [dict(unit=0, w=3, h=2, fmt="l", bytes=[1, 2, 3, 4, 5, 6])], unpack=1),
        gl_case("uniform sampler2D t0; void main() { gl_FragColor = texture2D(t0, uv); }", 1, 1,
                [dict(unit=0, w=1, h=1, fmt="la", bytes=[77, 200])]),
        # a unit without a complete texture samples as (0, 0, 0, 1)
        gl_case("uniform sampler2D t5; void main() { gl_FragColor = texture2D(t5, uv); }", 1, 1, [dict(unit=5, w=0, h=0, fmt="rgba", bytes=None)]),
        # UNORM8 -> float is c / 255 in binary32: 128 / 255 * 510 = 256.0000076..., which floors to 256
        gl_case("uniform sampler2D t0; void main() { float v = texture2D(t0, uv).r; gl_FragColor = vec4(floor(v * 510.0) / 1020.0, v, 0.0, 1.0); }", 1, 1,
                [dict(unit=0, w=1, h=1, fmt="l", bytes=[128])]),
        # varyings at pixel centres: uv.x = (i + 0.5) / 4
        gl_case("void main() { gl_FragColor = vec4(uv.x, uv.y, 0.0, 1.0); }", 4, 2),
        # D5: the store clamps, scales by 255 and rounds to nearest, ties to even: 0.5 -> 127.5 -> 128; 0.25 -> 63.75 -> 64;
        # 0.7f * 255 = 178.49999696 -> 178; 1.5 -> 255; -0.2 -> 0
        gl_case("void main() { gl_FragColor = vec4(0.5, 0.25, 0.7, 1.5); }", 1, 1),
        gl_case("void main() { gl_FragColor = vec4(-0.2, 0.498, 0.002, 0.00196); }", 1, 1),
        # the half-pel average the reference's shaders rely on: (a + b) / 2 / 255 + 1 / 512 stores as (a + b + 1) >> 1
        gl_case("void main() { gl_FragColor = vec4((7.0 + 8.0) / 510.0 + 1.0 / 512.0, (7.0 + 7.0) / 510.0 + 1.0 / 512.0, (255.0 + 254.0) / 510.0 + 1.0 / 512.0, (0.0 + 1.0) / 510.0 + 1.0 / 512.0); }", 1, 1),
    ]
    res = run(cases, tmp_path)
    assert not [r for r in res if "error" in r], [r for r in res if "error" in r]
    red = [r["bytes"][0] for r in res[:12]]
    # u * 4 snapped to 1/256 and floored: 0.249 * 4 = 0.996 -> 255/256 -> texel 0; 0.2499999 * 4 = 0.9999996 -> 256/256 -> texel 1 (D8);
    # coordinates beyond the edges clamp to the edge texel
    assert red == [10, 10, 10, 20, 20, 20, 30, 30, 40, 40, 40, 10], red
    assert res[12]["bytes"] == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    assert res[13]["bytes"] == [1, 1, 1, 255, 2, 2, 2, 255, 3, 3, 3, 255, 4, 4, 4, 255, 5, 5, 5, 255, 6, 6, 6, 255]
    assert res[14]["bytes"] == res[13]["bytes"]
    assert res[15]["bytes"] == [77, 77, 77, 200]
    assert res[16]["bytes"] == [0, 0, 0, 255]
    assert res[17]["bytes"] == [64, 128, 0, 255]          # floor(256.0000076) / 1020 * 255 = 64; 128 back out
    assert res[18]["bytes"][0::4] == [32, 96, 159, 223] * 2          # 0.125 * 255 = 31.875, 0.375 -> 95.625, 0.625 -> 159.375, 0.875 -> 223.125
    assert res[18]["bytes"][1::4] == [64] * 4 + [191] * 4            # uv.y = 0.25 / 0.75 -> 63.75 / 191.25; row 0 is the bottom row of the viewport
    assert res[19]["bytes"] == [128, 64, 178, 255]
    assert res[20]["bytes"] == [0, 127, 1, 0]              # 0.498 -> 126.99 -> 127; 0.002 -> 0.51 -> 1; 0.00196 -> 0.4998 -> 0
    assert res[21]["bytes"] == [8, 7, 255, 1]


def test_only_nearest_clamp_is_modelled(tmp_path):
    res = run([gl_case("uniform sampler2D t0; void main() { gl_FragColor = texture2D(t0, uv); }", 1, 1,
                       [dict(unit=0, w=1, h=1, fmt="rgba", bytes=[1, 2, 3, 4], filter="default")])], tmp_path)
    assert "error" in res[0] and "NEAREST" in res[0]["error"]
