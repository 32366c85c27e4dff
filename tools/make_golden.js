#!/usr/bin/env node
/*
 * make_golden.js -- generates tests/golden/*.json by EXECUTING the reference's own
 * JavaScript (unmodified, read from /root/reference at generation time) under Node.
 * Runs only in the build container; the resulting fixtures are data (inputs +
 * outputs) and are what travels.  No reference source text is written anywhere.
 *
 *   node tools/make_golden.js /root/reference tests/golden
 *
 * Fixtures:
 *   mc_copymacroblock.json  jsv.prototype.copyMacroblock (decoders/jsv.js:895-1129),
 *                           the CPU twin of the shader's predictor arithmetic
 *   rgb_ycbcrtorgba.json    jsv.prototype.YCbCrToRGBA (player/easybits.player.js:2674-2785)
 *   parser_*.json           boundary tensors T1-T5 recorded from the unmodified
 *                           reference parser + IDCT_GL (decoders/jsv.js:237-893,
 *                           :1177-1336) through a recording fake `gl`, for every
 *                           stream in tests/golden/streams/*.jsv
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const crypto = require('crypto');

const REF = process.argv[2] || '/root/reference';
const OUT = process.argv[3] || path.join(__dirname, '..', 'tests', 'golden');

function xorshift32(seed) {
  let s = seed >>> 0;
  return function () {
    s ^= s << 13; s >>>= 0;
    s ^= s >>> 17;
    s ^= s << 5; s >>>= 0;
    return s;
  };
}
const b64 = (typed) => Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength).toString('base64');

function loadDecoderContext() {
  const sandbox = {
    Uint8Array, Int16Array, Int32Array, Uint32Array, Uint8ClampedArray, Float32Array, ArrayBuffer,
    DataView, Math, Date, JSON, Object, Array, Error, parseInt, parseFloat, isNaN, NaN, Infinity,
    console: { log() {}, warn() {}, error() {}, info() {} },
    setTimeout, clearTimeout, setInterval, clearInterval,
    SHADER_VERTEX_IDENTITY: 'void main(){}',        // normally player/parts/end.js:158
    DEFAULT_SECONDS_PLAYED_LIMIT: 30,               // normally player/parts/end.js:65
  };
  sandbox.window = sandbox;
  vm.createContext(sandbox);
  for (const f of ['features/eventdispatcher.js', 'features/bitreader.js',
                   'decoders/shaders/mpeg1video.js', 'decoders/jsv.js']) {
    vm.runInContext(fs.readFileSync(path.join(REF, f), 'utf8'), sandbox, { filename: f });
  }
  return sandbox;
}

// ---------------------------------------------------------------- A. copyMacroblock
function goldenCopyMacroblock(ctx) {
  const proto = ctx.jsv_dec.prototype;
  const cw = 64, chh = 48, hw = cw >> 1, hh = chh >> 1;
  const rnd = xorshift32(0x4C454F4E);
  const sY = new Uint8Array(cw * chh), sCb = new Uint8Array(hw * hh), sCr = new Uint8Array(hw * hh);
  for (let i = 0; i < sY.length; i++) sY[i] = rnd() & 255;
  for (let i = 0; i < sCb.length; i++) { sCb[i] = rnd() & 255; sCr[i] = rnd() & 255; }
  const cases = [];
  for (const [mbRow, mbCol] of [[1, 1], [1, 2]]) {
    for (let mvV = -9; mvV <= 9; mvV++) {
      for (let mvH = -9; mvH <= 9; mvH++) {
        const dY = new Uint8Array(cw * chh), dCb = new Uint8Array(hw * hh), dCr = new Uint8Array(hw * hh);
        const self = {
          mbRow, mbCol, codedWidth: cw, halfWidth: hw,
          currentY32: new Uint32Array(dY.buffer), currentCb32: new Uint32Array(dCb.buffer),
          currentCr32: new Uint32Array(dCr.buffer),
        };
        proto.copyMacroblock.call(self, mvH, mvV, sY, sCr, sCb);   // note the (sY, sCr, sCb) order
        const oY = new Uint8Array(256), oCb = new Uint8Array(64), oCr = new Uint8Array(64);
        for (let r = 0; r < 16; r++) for (let c = 0; c < 16; c++) oY[r * 16 + c] = dY[(mbRow * 16 + r) * cw + mbCol * 16 + c];
        for (let r = 0; r < 8; r++) for (let c = 0; c < 8; c++) {
          oCb[r * 8 + c] = dCb[(mbRow * 8 + r) * hw + mbCol * 8 + c];
          oCr[r * 8 + c] = dCr[(mbRow * 8 + r) * hw + mbCol * 8 + c];
        }
        cases.push({ mbRow, mbCol, mvH, mvV, y: b64(oY), cb: b64(oCb), cr: b64(oCr) });
      }
    }
  }
  return { source: 'decoders/jsv.js:895-1129 jsv.prototype.copyMacroblock, executed under node ' + process.version,
           coded_w: cw, coded_h: chh, ref_y: b64(sY), ref_cb: b64(sCb), ref_cr: b64(sCr), cases };
}

// ---------------------------------------------------------------- B. YCbCrToRGBA
function goldenYCbCrToRGBA(ctx) {
  // take the function text out of the player file at generation time and evaluate it
  const src = fs.readFileSync(path.join(REF, 'player/easybits.player.js'), 'utf8');
  const start = src.indexOf('jsv.prototype.YCbCrToRGBA = function');
  const end = src.indexOf('jsv.prototype.renderFrameGL', start);
  if (start < 0 || end < 0) throw new Error('YCbCrToRGBA not found');
  const sandbox = { Uint8ClampedArray, window: {}, jsv: function () {} };
  vm.createContext(sandbox);
  vm.runInContext(src.slice(start, end), sandbox, { filename: 'player/easybits.player.js#YCbCrToRGBA' });
  const fn = sandbox.jsv.prototype.YCbCrToRGBA;
  const fillArray = ctx.jsv_dec.prototype.fillArray;
  const sets = [];
  function run(name, cw, chh, fw, fh, pY, pCb, pCr) {
    const self = {
      frameWidth: fw, frameHeight: fh, codedWidth: cw, halfWidth: cw >> 1,
      decoder: { fillArray },
      canvasContext: { createImageData: (w, h) => ({ data: new Uint8ClampedArray(w * h * 4), width: w, height: h }) },
    };
    const img = fn.call(self, pY, pCb, pCr);
    sets.push({ name, coded_w: cw, coded_h: chh, frame_w: fw, frame_h: fh,
                y: b64(pY), cb: b64(pCb), cr: b64(pCr), rgba: b64(new Uint8Array(img.data)) });
  }
  const rnd = xorshift32(0x52474241);
  for (const [name, cw, chh, fw, fh] of [['random_64x48_crop61x45', 64, 48, 61, 45], ['random_32x32_full', 32, 32, 32, 32]]) {
    const pY = new Uint8Array(cw * chh), pCb = new Uint8Array(cw * chh / 4), pCr = new Uint8Array(cw * chh / 4);
    for (let i = 0; i < pY.length; i++) pY[i] = rnd() & 255;
    for (let i = 0; i < pCb.length; i++) { pCb[i] = rnd() & 255; pCr[i] = rnd() & 255; }
    run(name, cw, chh, fw, fh, pY, pCb, pCr);
  }
  // exact decimal ties: coefficients have 5 decimals, so value*1e5 is an integer N and
  // N = 50000 (mod 100000) sits exactly on a rounding boundary -- only fp64 rounding
  // error decides the byte.  Collect such (y, cb, cr) triples per channel.
  const ties = [];
  const mod = (a, m) => ((a % m) + m) % m;
  for (let y = 0; y < 256; y++) for (let c = 0; c < 256; c++) {
    const nr = 159603 * (c - 128) + 116438 * (y - 16);
    const nb = 201723 * (c - 128) + 116438 * (y - 16);
    if (mod(nr, 100000) === 50000 && nr > 0 && nr < 25500000) ties.push([y, 128, c]);
    if (mod(nb, 100000) === 50000 && nb > 0 && nb < 25500000) ties.push([y, c, 128]);
  }
  for (let y = 0; y < 256; y += 1) for (let cb = 0; cb < 256; cb += 1) for (let cr = 0; cr < 256; cr += 1) {
    if (ties.length >= 2048) break;
    const ng = -81297 * (cr - 128) - 39176 * (cb - 128) + 116438 * (y - 16);
    if (mod(ng, 100000) === 50000 && ng > 0 && ng < 25500000) ties.push([y, cb, cr]);
  }
  {
    // one 2x2 quad per triple: coded width 64 -> 32 quads per quad-row
    const n = ties.length, cw = 64, quadRows = Math.ceil(n / 32), chh = Math.ceil(quadRows * 2 / 16) * 16;
    const pY = new Uint8Array(cw * chh), pCb = new Uint8Array(cw * chh / 4).fill(128), pCr = new Uint8Array(cw * chh / 4).fill(128);
    ties.forEach(([y, cb, cr], i) => {
      const qx = i % 32, qy = (i / 32) | 0;
      pCb[qy * 32 + qx] = cb; pCr[qy * 32 + qx] = cr;
      for (let dy = 0; dy < 2; dy++) for (let dx = 0; dx < 2; dx++) pY[(2 * qy + dy) * cw + 2 * qx + dx] = y;
    });
    run('decimal_ties', cw, chh, cw, chh, pY, pCb, pCr);
  }
  return { source: 'player/easybits.player.js:2674-2785 jsv.prototype.YCbCrToRGBA, executed under node ' + process.version,
           n_ties: ties.length, sets };
}

// ---------------------------------------------------------------- C. parser boundary tensors
function recordingGL(rec) {
  let boundUnit = 0, fbo = null;
  const handler = {
    get(_t, name) {
      if (name === 'getShaderPrecisionFormat') return () => ({ rangeMin: 30, rangeMax: 30, precision: 0 });
      if (name === 'getProgramParameter' || name === 'getShaderParameter') return () => true;
      if (name === 'createTexture') return () => ({ kind: 'tex' });
      if (name === 'createFramebuffer') return () => ({ kind: 'fbo' });
      if (name === 'createProgram') return () => ({ kind: 'prog' });
      if (name === 'createShader') return () => ({ kind: 'shader' });
      if (name === 'getUniformLocation') return (_p, n) => ({ uniform: n });
      if (name === 'getAttribLocation') return () => 0;
      if (name === 'activeTexture') return (u) => { boundUnit = u; };
      if (name === 'bindFramebuffer') return (_t2, f) => { fbo = f; };
      if (name === 'uniform1f') return (loc, v) => { if (loc && loc.uniform === '_ac') rec.push({ op: 'mv_coef', v }); };
      if (name === 'texImage2D') return (...a) => {
        const data = a[8];
        if (data && data.length !== undefined)
          rec.push({ op: 'tex', unit: boundUnit - 0x84C0, w: a[3], h: a[4], fmt: a[6], data: new Uint8Array(data.buffer ? data.buffer.slice(data.byteOffset, data.byteOffset + data.byteLength) : data) });
      };
      if (name === 'drawArrays') return () => rec.push({ op: 'draw' });
      if (typeof name === 'string' && /^[A-Z0-9_]+$/.test(name)) {
        const K = { TEXTURE0: 0x84C0, TEXTURE1: 0x84C1, TEXTURE2: 0x84C2, TEXTURE3: 0x84C3, TEXTURE4: 0x84C4,
                    TEXTURE5: 0x84C5, TEXTURE6: 0x84C6, TEXTURE7: 0x84C7, LUMINANCE: 0x1909, LUMINANCE_ALPHA: 0x190A, RGBA: 0x1908 };
        return name in K ? K[name] : 1;
      }
      return () => undefined;
    },
  };
  return new Proxy({}, handler);
}

function goldenParser(ctx, streamPath) {
  const bytes = new Uint8Array(fs.readFileSync(streamPath));
  const dec = new ctx.jsv_dec();
  const rec = [];
  dec._initGL(recordingGL(rec));
  const events = [];
  const pictures = [];
  let cur = null;
  dec.on('meta', (e) => { const m = e.detail; events.push({ ev: 'meta', w: m.w, h: m.h, d: m.d, a: m.a }); });
  dec.on('seq', (e) => { const s = e.detail; events.push({ ev: 'seq', r: s.r, w: s.w, h: s.h }); });
  dec.on('ended', () => events.push({ ev: 'ended' }));
  dec.on('frame', (e) => {
    const f = e.detail;
    f.ybr[0].inuse = 0;                      // stands in for renderFrameGL (player.js:2820)
    // collect the uploads of this picture from the recording
    const pic = { ts: f.ts, type: dec.pictureCodingType, uploads: [] };
    for (const r of rec.splice(0)) {
      if (r.op === 'tex') {
        // big arrays travel as a digest, small ones in full
        const u = { unit: r.unit, w: r.w, h: r.h, fmt: r.fmt, bytes: r.data.length,
                    sha256: crypto.createHash('sha256').update(r.data).digest('hex') };
        if (r.data.length <= 4096) u.data = b64(r.data);
        pic.uploads.push(u);
      }
      else if (r.op === 'mv_coef') pic.uploads.push({ mv_coef: r.v });
    }
    pictures.push(pic);
    events.push({ ev: 'frame', ts: f.ts });
  });
  dec.buffer.addBuffer({ data: bytes, start: 0, end: bytes.length - 1, total: bytes.length });
  dec._initMeta();
  dec._skipTillGop = true;
  rec.splice(0);                             // drop the constant-table uploads of _initGL
  const ended = () => events.length > 0 && events[events.length - 1].ev === 'ended';
  for (let i = 0; i < 100000 && !ended(); i++) {
    const before = events.length;
    dec.decodeFrame();
    if (events.length === before && i > 50000) break;
  }
  return { source: 'decoders/jsv.js parser + IDCT_GL uploads, unmodified, node ' + process.version,
           stream: path.basename(streamPath), mbWidth: dec.mbWidth, mbHeight: dec.mbHeight,
           codedWidth: dec.codedWidth, codedHeight: dec.codedHeight, events, pictures };
}

function main() {
  fs.mkdirSync(OUT, { recursive: true });
  const ctx = loadDecoderContext();
  const what = process.argv[4] || 'all';
  if (what === 'all' || what === 'mc')
    fs.writeFileSync(path.join(OUT, 'mc_copymacroblock.json'), JSON.stringify(goldenCopyMacroblock(ctx)));
  if (what === 'all' || what === 'rgb')
    fs.writeFileSync(path.join(OUT, 'rgb_ycbcrtorgba.json'), JSON.stringify(goldenYCbCrToRGBA(ctx)));
  if (what === 'all' || what === 'parser') {
    const sdir = path.join(OUT, 'streams');
    if (fs.existsSync(sdir))
      for (const f of fs.readdirSync(sdir).filter((x) => x.endsWith('.jsv')).sort()) {
        const g = goldenParser(loadDecoderContext(), path.join(sdir, f));
        fs.writeFileSync(path.join(OUT, 'parser_' + f.replace(/\.jsv$/, '') + '.json'), JSON.stringify(g));
        console.log(f, '->', g.pictures.length, 'pictures', g.events[g.events.length - 1]);
      }
  }
}
main();
