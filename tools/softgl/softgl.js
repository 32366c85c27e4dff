/*
 * softgl.js -- a software WebGL-1 context, just large enough to run the reference's UNMODIFIED GL
 * driver code (decoders/jsv.js:51-236 set-up, :1177-1336 IDCT_GL; player/easybits.player.js
 * :2787-2858 renderFrameGL, :2900-2944 initWebGL) with its UNMODIFIED shader text executed by
 * tools/softgl/glsl.js.  TEST TOOLING, build container only: it turns "WebGL cannot run here" into
 * "the reference's own pixel path runs here, slowly", which is what pins the oracle's
 * dequantisation + IDCT (SURVEY.md 8c).  Nothing in here knows what the shaders compute.
 *
 * The GL machine that is modelled (OpenGL ES 2.0 full specification sections in brackets):
 *   textures      RGBA / LUMINANCE / LUMINANCE_ALPHA, UNSIGNED_BYTE, UNPACK_ALIGNMENT honoured
 *                 [3.7.1, table 3.12: L -> (L,L,L,1), LA -> (L,L,L,A)]; UNORM8 -> float = c/255
 *                 rounded to binary32 [2.1.2]
 *   sampling      NEAREST + CLAMP_TO_EDGE only (what the reference sets, jsv.js:213-217):
 *                 texel i = floor(u * width) clamped [3.7.7], with u * width first snapped to
 *                 8 fractional bits, round to nearest -- the sub-texel precision of D3D10+/GCN-class
 *                 samplers.  This is decision D8 of SURVEY.md 8c stated as a property of the
 *                 sampler instead of per call site: a coordinate that the shader computes to land
 *                 exactly on a texel edge, and that binary32 rounding leaves a few ulp short of it,
 *                 reads the texel that starts at that edge.
 *   rasterisation TRIANGLE_STRIP of 4 vertices that form an axis-aligned rectangle; one fragment per
 *                 pixel centre inside the viewport [3.5.1]; varyings interpolated in double
 *                 precision from the vertex shader's outputs and rounded to binary32
 *   output        colour attachment RGBA8: clamp to [0,1], * 255, round to nearest even [2.1.2, 4.2.x]
 *   not modelled  blending, depth, stencil, mipmaps, scissor, dithering (all off in the reference)
 */
'use strict';
const glsl = require('./glsl.js');

const K = {
  TEXTURE_2D: 0x0DE1, TEXTURE0: 0x84C0, RGBA: 0x1908, LUMINANCE: 0x1909, LUMINANCE_ALPHA: 0x190A, UNSIGNED_BYTE: 0x1401,
  NEAREST: 0x2600, LINEAR: 0x2601, CLAMP_TO_EDGE: 0x812F, REPEAT: 0x2901,
  TEXTURE_MAG_FILTER: 0x2800, TEXTURE_MIN_FILTER: 0x2801, TEXTURE_WRAP_S: 0x2802, TEXTURE_WRAP_T: 0x2803,
  FRAMEBUFFER: 0x8D40, COLOR_ATTACHMENT0: 0x8CE0, ARRAY_BUFFER: 0x8892, STATIC_DRAW: 0x88E4, FLOAT: 0x1406,
  VERTEX_SHADER: 0x8B31, FRAGMENT_SHADER: 0x8B30, COMPILE_STATUS: 0x8B81, LINK_STATUS: 0x8B82,
  TRIANGLE_STRIP: 0x0005, UNPACK_ALIGNMENT: 0x0CF5, UNPACK_FLIP_Y_WEBGL: 0x9240,
  HIGH_FLOAT: 0x8DF2, MEDIUM_FLOAT: 0x8DF1, LOW_FLOAT: 0x8DF0, HIGH_INT: 0x8DF5, MEDIUM_INT: 0x8DF4, LOW_INT: 0x8DF3,
  COLOR_BUFFER_BIT: 0x4000,
};
for (let i = 1; i < 16; i++) K['TEXTURE' + i] = K.TEXTURE0 + i;

const fr = Math.fround;
const U8_TO_F = new Float32Array(256);
for (let i = 0; i < 256; i++) U8_TO_F[i] = fr(i / 255);

function roundHalfEven(x) {
  const f = Math.floor(x), d = x - f;
  if (d > 0.5) return f + 1;
  if (d < 0.5) return f;
  return f % 2 === 0 ? f : f + 1;
}

function createContext(opts) {
  opts = opts || {};
  const canvasW = opts.width || 0, canvasH = opts.height || 0;
  const st = {
    unit: 0, units: new Array(16).fill(null), arrayBuffer: null, attrib: null, program: null, fbo: null,
    viewport: [0, 0, canvasW, canvasH], unpackAlign: 4,
    canvas: { w: canvasW, h: canvasH, data: new Uint8Array(canvasW * canvasH * 4) },
    draws: 0, fragments: 0, incompleteFetches: 0,
  };
  const curTex = () => { const t = st.units[st.unit]; if (!t) throw new Error('softgl: no texture bound on unit ' + st.unit); return t; };

  // texture2D for glsl.js: the sampler uniform holds a texture unit number
  function fetch(unit, u, v) {
    const t = st.units[unit];
    // no texture object bound, or an incomplete one: the sampler returns (0, 0, 0, 1) [3.8.2] -- the
    // reference relies on this for its unit 5 (the bind is commented out, decoders/jsv.js:1219-1221)
    if (!t || !t.data || t.w === 0 || t.h === 0) { st.incompleteFetches++; return [0, 0, 0, 1]; }
    if (t.mag !== K.NEAREST || t.min !== K.NEAREST || t.wrapS !== K.CLAMP_TO_EDGE || t.wrapT !== K.CLAMP_TO_EDGE)
      throw new Error('softgl: only NEAREST + CLAMP_TO_EDGE sampling is modelled');
    // scaled coordinate -> 8 fractional bits (round to nearest) -> floor -> clamp
    const SN = process.env.SOFTGL_NO_SNAP ? 0 : 1; let x = SN ? Math.floor(Math.floor(u * t.w * 256 + 0.5) / 256) : Math.floor(u * t.w), y = SN ? Math.floor(Math.floor(v * t.h * 256 + 0.5) / 256) : Math.floor(v * t.h);
    x = x < 0 ? 0 : x >= t.w ? t.w - 1 : x;
    y = y < 0 ? 0 : y >= t.h ? t.h - 1 : y;
    const o = (y * t.w + x) * 4, d = t.data;
    return [U8_TO_F[d[o]], U8_TO_F[d[o + 1]], U8_TO_F[d[o + 2]], U8_TO_F[d[o + 3]]];
  }

  const gl = Object.assign({}, K, {
    _state: st,
    getShaderPrecisionFormat(_shaderType, prec) {
      // a desktop-class implementation: binary32 floats, 32-bit two's complement ints
      if (prec === K.HIGH_INT || prec === K.MEDIUM_INT || prec === K.LOW_INT) return { rangeMin: 31, rangeMax: 30, precision: 0 };
      return { rangeMin: 127, rangeMax: 127, precision: 23 };
    },
    pixelStorei(p, v) { if (p === K.UNPACK_ALIGNMENT) st.unpackAlign = v; else if (p === K.UNPACK_FLIP_Y_WEBGL && v) throw new Error('softgl: UNPACK_FLIP_Y not modelled'); },
    createTexture() { return { kind: 'tex', w: 0, h: 0, data: null, mag: K.LINEAR, min: 0x2702, wrapS: K.REPEAT, wrapT: K.REPEAT }; },
    createFramebuffer() { return { kind: 'fbo', tex: null }; },
    createBuffer() { return { kind: 'buf', data: null }; },
    createProgram() { return { kind: 'prog', shaders: [], uniforms: new Map(), linked: false }; },
    createShader(type) { return { kind: 'shader', type, src: null, ok: false, log: '' }; },
    shaderSource(s, src) { s.src = src; },
    compileShader(s) {
      try { s.prog = glsl.compile(s.src, fetch); s.ok = true; } catch (e) { s.ok = false; s.log = String(e.message); }
    },
    getShaderParameter(s, p) { return p === K.COMPILE_STATUS ? s.ok : null; },
    getShaderInfoLog(s) { return s.log; },
    attachShader(p, s) { p.shaders.push(s); },
    linkProgram(p) {
      p.vs = p.shaders.find((s) => s.type === K.VERTEX_SHADER);
      p.fs = p.shaders.find((s) => s.type === K.FRAGMENT_SHADER);
      p.linked = !!(p.vs && p.fs && p.vs.ok && p.fs.ok);
      if (!p.linked) return;
      p.vG = p.vs.prog.instantiate();
      p.fG = p.fs.prog.instantiate();
      // varyings are matched by name
      p.varyings = [];
      for (const [name, g] of p.vs.prog.globals) if (g.qual === 'varying') {
        const f = p.fs.prog.globals.get(name);
        if (f && f.qual === 'varying') p.varyings.push({ name, vslot: g.slot, fslot: f.slot, type: g.type });
      }
    },
    getProgramParameter(p, q) { return q === K.LINK_STATUS ? p.linked : null; },
    getProgramInfoLog(p) { return (p.vs ? p.vs.log : 'no vertex shader') + ' ' + (p.fs ? p.fs.log : 'no fragment shader'); },
    useProgram(p) { st.program = p; },
    getAttribLocation(p, name) { const g = p.vs.prog.globals.get(name); return g && g.qual === 'attribute' ? g.slot : -1; },
    enableVertexAttribArray() {},
    vertexAttribPointer(loc, size, type, _norm, stride, offset) {
      if (type !== K.FLOAT || stride !== 0 || offset !== 0) throw new Error('softgl: only tightly packed float attributes');
      st.attrib = { loc, size, buffer: st.arrayBuffer };
    },
    bindBuffer(_t, b) { st.arrayBuffer = b; },
    bufferData(_t, data) { st.arrayBuffer.data = new Float32Array(data); },
    getUniformLocation(p, name) {
      const out = [];
      for (const sh of [p.vs, p.fs]) { const g = sh.prog.globals.get(name); if (g && g.qual === 'uniform') out.push({ G: sh === p.vs ? p.vG : p.fG, slot: g.slot, type: g.type }); }
      return out.length ? { prog: p, name, where: out } : null;
    },
    uniform1i(loc, v) {
      if (!loc) return;
      if (loc.prog !== st.program) throw new Error('softgl: uniform1i on a program that is not current');
      for (const w of loc.where) { if (w.type !== 'int' && w.type !== 'sampler2D') throw new Error('softgl: uniform1i on ' + w.type + ' ' + loc.name); w.G[w.slot] = v | 0; }
    },
    uniform1f(loc, v) {
      if (!loc) return;
      if (loc.prog !== st.program) throw new Error('softgl: uniform1f on a program that is not current');
      for (const w of loc.where) { if (w.type !== 'float') throw new Error('softgl: uniform1f on ' + w.type + ' ' + loc.name); w.G[w.slot] = fr(v); }
    },
    activeTexture(u) { st.unit = u - K.TEXTURE0; },
    bindTexture(_t, tex) { st.units[st.unit] = tex; },
    texParameteri(_t, p, v) {
      const t = curTex();
      if (p === K.TEXTURE_MAG_FILTER) t.mag = v; else if (p === K.TEXTURE_MIN_FILTER) t.min = v;
      else if (p === K.TEXTURE_WRAP_S) t.wrapS = v; else if (p === K.TEXTURE_WRAP_T) t.wrapT = v;
    },
    texImage2D(_t, level, ifmt, w, h, border, fmt, type, data) {
      if (level !== 0 || border !== 0 || type !== K.UNSIGNED_BYTE || ifmt !== fmt) throw new Error('softgl: texImage2D form not modelled');
      if (!Number.isInteger(w) || !Number.isInteger(h) || w < 0 || h < 0) throw new Error('softgl: texture size ' + w + 'x' + h);
      const t = curTex();
      const comps = fmt === K.RGBA ? 4 : fmt === K.LUMINANCE_ALPHA ? 2 : fmt === K.LUMINANCE ? 1 : 0;
      if (!comps) throw new Error('softgl: texture format 0x' + fmt.toString(16));
      t.w = w; t.h = h; t.fmt = fmt;
      t.data = new Uint8Array(w * h * 4);
      if (data == null) return;
      const src = data instanceof Uint8Array ? data : new Uint8Array(data.buffer, data.byteOffset, data.byteLength);
      const rowBytes = w * comps, stride = Math.ceil(rowBytes / st.unpackAlign) * st.unpackAlign;
      if (src.length < stride * (h - 1) + rowBytes) throw new Error('softgl: texImage2D source too small: ' + src.length + ' bytes for ' + w + 'x' + h + 'x' + comps);
      for (let y = 0; y < h; y++) for (let x = 0; x < w; x++) {
        const s = y * stride + x * comps, o = (y * w + x) * 4;
        if (comps === 4) { t.data[o] = src[s]; t.data[o + 1] = src[s + 1]; t.data[o + 2] = src[s + 2]; t.data[o + 3] = src[s + 3]; }
        else { t.data[o] = t.data[o + 1] = t.data[o + 2] = src[s]; t.data[o + 3] = comps === 2 ? src[s + 1] : 255; }
      }
    },
    bindFramebuffer(_t, f) { st.fbo = f; },
    framebufferTexture2D(_t, _att, _tt, tex) { st.fbo.tex = tex; },
    viewport(x, y, w, h) { st.viewport = [x, y, w, h]; },
    clearColor() {}, clear() {}, disable() {}, enable() {}, flush() {}, finish() {},

    drawArrays(mode, first, count) {
      if (mode !== K.TRIANGLE_STRIP || first !== 0 || count !== 4) throw new Error('softgl: only drawArrays(TRIANGLE_STRIP, 0, 4)');
      const p = st.program;
      if (!p || !p.linked) throw new Error('softgl: no linked program');
      const target = st.fbo ? st.fbo.tex : st.canvas;
      if (!target || !target.data) throw new Error('softgl: incomplete framebuffer');
      if (st.fbo) for (const t of st.units) if (t === target && t) {
        // a feedback loop is undefined behaviour in GL; the reference never creates one
        for (const [, g] of p.fs.prog.globals) if (g.type === 'sampler2D' && st.units[p.fG[g.slot]] === target) throw new Error('softgl: render target is also sampled');
      }
      // ---- vertex stage
      const a = st.attrib;
      if (!a || !a.buffer || !a.buffer.data || a.size !== 2) throw new Error('softgl: vertex attribute not set up');
      const verts = [];
      for (let i = 0; i < 4; i++) {
        p.vG[a.loc] = [fr(a.buffer.data[2 * i]), fr(a.buffer.data[2 * i + 1])];
        p.vs.prog.main(p.vG);
        const pos = p.vG[p.vs.prog.globals.get('gl_Position').slot];
        verts.push({ x: pos[0] / pos[3], y: pos[1] / pos[3], vary: p.varyings.map((v) => p.vG[v.vslot].slice()) });
      }
      // window coordinates of the vertices [2.12.1]; the strip must be a parallelogram:
      // w = w0 + s * (w2 - w0) + t * (w1 - w0) with w3 = w1 + w2 - w0
      const [vx, vy, vw, vh] = st.viewport;
      for (const v of verts) { v.wx = (v.x + 1) * vw / 2; v.wy = (v.y + 1) * vh / 2; }
      const [v0, v1, v2, v3] = verts;
      if (Math.abs(v3.wx - (v1.wx + v2.wx - v0.wx)) > 1e-9 || Math.abs(v3.wy - (v1.wy + v2.wy - v0.wy)) > 1e-9) throw new Error('softgl: strip is not a parallelogram');
      const ex = { x: v2.wx - v0.wx, y: v2.wy - v0.wy }, ey = { x: v1.wx - v0.wx, y: v1.wy - v0.wy };
      const det = ex.x * ey.y - ex.y * ey.x;
      if (det === 0) throw new Error('softgl: degenerate quad');
      // ---- fragment stage
      const fragSlot = p.fs.prog.globals.get('gl_FragColor').slot;
      const G = p.fG;
      for (let j = 0; j < vh; j++) {
        const wy = vy + j;
        if (wy < 0 || wy >= target.h) continue;
        for (let i = 0; i < vw; i++) {
          const wx = vx + i;
          if (wx < 0 || wx >= target.w) continue;
          // pixel centre (i + 0.5, j + 0.5) in viewport-relative window coordinates
          const dx = i + 0.5 - v0.wx, dy = j + 0.5 - v0.wy;
          const s = (dx * ey.y - dy * ey.x) / det, t = (ex.x * dy - ex.y * dx) / det;
          if (s < 0 || s > 1 || t < 0 || t > 1) continue;
          for (let k = 0; k < p.varyings.length; k++) {
            const va = p.varyings[k], a0 = v0.vary[k], a1 = v1.vary[k], a2 = v2.vary[k];
            G[va.fslot] = a0.map((c, q) => fr(c + s * (a2[q] - c) + t * (a1[q] - c)));
          }
          G[fragSlot] = [0, 0, 0, 0];
          p.fs.prog.main(G);
          const c = G[fragSlot], o = (wy * target.w + wx) * 4;
          for (let q = 0; q < 4; q++) {
            const f = c[q] < 0 ? 0 : c[q] > 1 ? 1 : c[q];
            if (f !== f) throw new Error('softgl: NaN colour at fragment ' + i + ',' + j);
            target.data[o + q] = roundHalfEven(f * 255);
          }
          st.fragments++;
        }
      }
      st.draws++;
    },

    // read a texture back as it is stored: row 0 first (tools only; WebGL has readPixels on FBOs)
    _textureBytes(tex) { return new Uint8Array(tex.data); },
    _canvasBytes() { return new Uint8Array(st.canvas.data); },
  });
  return gl;
}

module.exports = { createContext, K };
