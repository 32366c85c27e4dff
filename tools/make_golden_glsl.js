#!/usr/bin/env node
/*
 * make_golden_glsl.js -- golden vectors for dequantisation + IDCT (+ forward MC) produced by
 * EXECUTING the reference's own pixel path: its unmodified parser and GL driver
 * (decoders/jsv.js), its unmodified shader text as composed by its own composeShaders()
 * (decoders/jsv.js:2459-2470 over decoders/shaders/mpeg1video.js:18-29), and the vertex /
 * colour shaders of player/parts/end.js:77-166, on the software WebGL-1 machine of
 * tools/softgl/ (a GLSL ES 1.00 evaluator + the GL state machine; no shader arithmetic is
 * typed anywhere on this path).  Build container only; reads /root/reference at generation
 * time; only the resulting DATA (inputs + outputs) is written.
 *
 *   python3 tools/make_glsl_cases.py /tmp/glsl_cases.json     # inputs (seeded, deterministic)
 *   node tools/make_golden_glsl.js /root/reference tests/golden /tmp/glsl_cases.json
 *
 * Writes
 *   tests/golden/glsl_idct_cases.json   picture sequences driven through jsv.prototype.IDCT_GL
 *                                       (decoders/jsv.js:1177-1336) with given boundary tensors:
 *                                       inputs, pass-1 scratch textures, output planes
 *   tests/golden/glsl_streams.json      whole fixture streams through the reference's
 *                                       decodeFrame loop: per picture the tensors it uploaded
 *                                       (digest), the output planes (full for small streams,
 *                                       digest for 352x240) and the canvas RGBA of renderFrameGL
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const zlib = require('zlib');
const crypto = require('crypto');
const softgl = require('./softgl/softgl.js');

const REF = process.argv[2] || '/root/reference';
const OUT = process.argv[3] || path.join(__dirname, '..', 'tests', 'golden');
const CASES = process.argv[4];

const zb64 = (u8) => zlib.deflateSync(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength), { level: 9 }).toString('base64');
const unz = (s, Ctor) => { const b = zlib.inflateSync(Buffer.from(s, 'base64')); const a = new Uint8Array(b.length); a.set(b); return Ctor ? new Ctor(a.buffer) : a; };
const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex');

// the shader strings of player/parts/end.js are array literals joined with '\n': evaluate the literal
function endJsString(name) {
  const src = fs.readFileSync(path.join(REF, 'player/parts/end.js'), 'utf8');
  const at = src.indexOf(name + ' = [');
  if (at < 0) throw new Error(name + ' not found in player/parts/end.js');
  const open = src.indexOf('[', at), close = src.indexOf("].join('\\n')", open);
  if (close < 0) throw new Error('end of ' + name + ' not found');
  return vm.runInNewContext('(' + src.slice(open, close + 1) + ")", {}).join('\n');
}

function loadReference() {
  const sandbox = {
    Uint8Array, Int16Array, Int32Array, Uint32Array, Uint8ClampedArray, Float32Array, ArrayBuffer,
    DataView, Math, Date, JSON, Object, Array, Error, parseInt, parseFloat, isNaN, NaN, Infinity,
    console: { log() {}, warn() {}, error() {}, info() {} },
    setTimeout, clearTimeout, setInterval, clearInterval,
    SHADER_VERTEX_IDENTITY: endJsString('SHADER_VERTEX_IDENTITY'),            // player/parts/end.js:158-166
    SHADER_FRAGMENT_YCBCRTORGBA: endJsString('SHADER_FRAGMENT_YCBCRTORGBA'),  // player/parts/end.js:77-156
    DEFAULT_SECONDS_PLAYED_LIMIT: 30,                                         // player/parts/end.js:65
  };
  sandbox.window = sandbox;
  vm.createContext(sandbox);
  for (const f of ['features/eventdispatcher.js', 'features/bitreader.js', 'decoders/shaders/mpeg1video.js', 'decoders/jsv.js'])
    vm.runInContext(fs.readFileSync(path.join(REF, f), 'utf8'), sandbox, { filename: f });
  // the player's GL half (createTexture, compileShader, initWebGL, renderFrameGL), as it stands in
  // player/easybits.player.js:2787-2944, attached to a bare constructor
  const psrc = fs.readFileSync(path.join(REF, 'player/easybits.player.js'), 'utf8');
  const a = psrc.indexOf('jsv.prototype.renderFrameGL = function');
  const b0 = psrc.indexOf('jsv.prototype.initWebGL = function', a);
  const b = psrc.indexOf('\n};', b0);
  if (a < 0 || b0 < 0 || b < 0) throw new Error('player GL functions not found');
  vm.runInContext('var jsv = function(){};\n' + psrc.slice(a, b + 3) + '\nwindow.player_ctor = jsv;', sandbox, { filename: 'player/easybits.player.js#gl' });
  return sandbox;
}

// one reference decoder wired the way the player wires it (player/easybits.player.js:584-585)
function makeDecoder(ctx, frameW, frameH) {
  const gl = softgl.createContext({ width: frameW, height: frameH });
  const player = new ctx.player_ctor();
  player.canvasEl = { getContext: () => gl };
  if (!player.initWebGL()) throw new Error('initWebGL failed');
  const dec = new ctx.jsv_dec();
  dec._initGL(gl);
  if (!dec.integer) throw new Error('the reference chose the float-emulation flavour');
  return { gl, player, dec };
}

function texBytes(gl, tex) { return gl._textureBytes(tex); }

function grabPicture(R, frameTextures) {
  const { gl, dec } = R;
  return {
    scratch: [0, 1, 2].map((c) => texBytes(gl, dec.glFrameBuffers.idct_1d[c].texture)),
    planes: frameTextures.map((t) => texBytes(gl, t)),
  };
}

function tensorsOf(dec) {
  const t = {
    type: dec.pictureCodingType,
    coef_y: new Int16Array(dec.currentYDCT16), coef_cb: new Int16Array(dec.currentCbDCT16), coef_cr: new Int16Array(dec.currentCrDCT16),
    qscale: new Uint8Array(dec.macroblockQuant), intra: new Uint8Array(dec.macroblockIsIntra),
  };
  if (dec.pictureCodingType === 2) { t.repadd = new Uint8Array(dec.macroblockRepAdd); t.mv_fwd = new Int16Array(dec.macroblockMV); }
  return t;
}

// decode a whole stream through the reference's own decodeFrame loop
function runStream(ctx, streamPath, onPicture) {
  const bytes = new Uint8Array(fs.readFileSync(streamPath));
  const w = (bytes[2] << 8) | bytes[3], h = (bytes[4] << 8) | bytes[5];
  const R = makeDecoder(ctx, w, h);
  const { dec } = R;
  let ended = false, held = null, n = 0;
  dec.on('ended', () => { ended = true; });
  dec.on('frame', (e) => {
    const f = e.detail;
    onPicture(R, f, n++);
    // the slot of the previous picture is released only now (renderFrameGL does it on display,
    // player/easybits.player.js:2820); releasing the newest one at once would let the next P
    // picture render into its own reference
    if (held) held.inuse = 0;
    held = f.ybr[0];
  });
  dec.buffer.addBuffer({ data: bytes, start: 0, end: bytes.length - 1, total: bytes.length });
  dec._initMeta();
  dec._skipTillGop = true;
  for (let i = 0; i < 200000 && !ended; i++) dec.decodeFrame();
  if (!ended) throw new Error('stream did not end: ' + streamPath);
  return R;
}

function main() {
  fs.mkdirSync(OUT, { recursive: true });
  const sdir = path.join(OUT, 'streams');
  const t0 = Date.now();

  // ---------------------------------------------------------------- streams
  const streams = [];
  for (const f of fs.readdirSync(sdir).filter((x) => x.endsWith('.jsv')).sort()) {
    if (/ibbp/.test(f)) continue;                 // the reference drops B pictures (decoders/jsv.js:613-616)
    const ctx = loadReference();
    const pics = [];
    const full = fs.statSync(path.join(sdir, f)).size < 40000;
    const R = runStream(ctx, path.join(sdir, f), (R2, frame, n) => {
      const g = grabPicture(R2, frame.ybr);
      const t = tensorsOf(R2.dec);
      const rec = { type: t.type, ts: frame.ts,
                    tensors_sha256: Object.fromEntries(Object.entries(t).filter(([, v]) => v.buffer).map(([k, v]) => [k, sha(v)])),
                    planes_sha256: g.planes.map(sha), scratch_sha256: g.scratch.map(sha) };
      if (full) { rec.planes = g.planes.map(zb64); }
      if (full || n < 2) {
        // the display conversion of the same picture: renderFrameGL onto the frame-sized canvas
        R2.player.codedWidth = R2.dec.codedWidth; R2.player.codedHeight = R2.dec.codedHeight;
        R2.player.frameWidth = R2.dec.frameWidth; R2.player.frameHeight = R2.dec.frameHeight;
        const inuse = frame.ybr[0].inuse;
        R2.player.renderFrameGL(frame);
        frame.ybr[0].inuse = inuse;               // the harness releases slots itself (see runStream)
        const canvas = R2.gl._canvasBytes();
        rec.canvas_sha256 = sha(canvas);
        if (full) rec.canvas = zb64(canvas);
      }
      pics.push(rec);
    });
    streams.push({ stream: f, coded_w: R.dec.codedWidth, coded_h: R.dec.codedHeight, frame_w: R.dec.frameWidth, frame_h: R.dec.frameHeight,
                   custom_intra_matrix: zb64(new Uint8Array(R.dec.QUANT_MATRIX)), pictures: pics });
    console.log(f, '->', pics.length, 'pictures,', R.gl._state.fragments, 'fragments,', ((Date.now() - t0) / 1000).toFixed(1), 's');
  }
  fs.writeFileSync(path.join(OUT, 'glsl_streams.json'), JSON.stringify({
    source: 'unmodified decoders/jsv.js (parser, IDCT_GL) + composeShaders() text + player/parts/end.js shaders + player renderFrameGL, ' +
            'executed on tools/softgl under node ' + process.version,
    canvas_note: 'canvas = RGBA8 rows bottom-up (GL window order) of the frame_w x frame_h default framebuffer',
    streams }));

  // ---------------------------------------------------------------- tensor-driven cases
  if (CASES) {
    const cases = JSON.parse(fs.readFileSync(CASES, 'utf8'));
    const out = [];
    for (const c of cases) {
      // the stream initialises the decoder (sequence header: size, matrices) the reference's own way
      const ctx = loadReference();
      const R = runStream(ctx, path.join(sdir, c.stream), () => {});
      const { dec } = R;
      const pics = [];
      let held = null;
      for (const p of c.pictures) {
        // what decodePicture does around its IDCT_GL call (decoders/jsv.js:619-622, :639-649, :662-665),
        // with the slice loop's results replaced by the given tensors
        dec.pictureCodingType = p.type;
        dec.currentYDCT16 = unz(p.coef_y, Int16Array); dec.currentYDCTU8 = new Uint8Array(dec.currentYDCT16.buffer);
        dec.currentCbDCT16 = unz(p.coef_cb, Int16Array); dec.currentCbDCTU8 = new Uint8Array(dec.currentCbDCT16.buffer);
        dec.currentCrDCT16 = unz(p.coef_cr, Int16Array); dec.currentCrDCTU8 = new Uint8Array(dec.currentCrDCT16.buffer);
        dec.macroblockQuant = unz(p.qscale); dec.macroblockIsIntra = unz(p.intra);
        if (p.type === 2) {
          dec.macroblockMV = unz(p.mv_fwd, Int16Array); dec.macroblockMVUint8 = new Uint8Array(dec.macroblockMV.buffer);
          dec.macroblockRepAdd = unz(p.repadd);
        }
        dec.IDCT_GL();
        dec.prev_pic_framebuffer = dec.framebuffer;
        const g = grabPicture(R, dec.framebuffer.map((x) => x.texture));
        if (held) held.inuse = 0;
        held = dec.framebuffer[0].texture;
        pics.push(Object.assign({}, p, { scratch: g.scratch.map(zb64), planes: g.planes.map(zb64) }));
      }
      out.push({ name: c.name, stream: c.stream, coded_w: dec.codedWidth, coded_h: dec.codedHeight,
                 quant_matrices: zb64(new Uint8Array(dec.QUANT_MATRIX)), pictures: pics });
      console.log('case', c.name, '->', pics.length, 'pictures,', ((Date.now() - t0) / 1000).toFixed(1), 's');
    }
    fs.writeFileSync(path.join(OUT, 'glsl_idct_cases.json'), JSON.stringify({
      source: 'jsv.prototype.IDCT_GL (decoders/jsv.js:1177-1336) + composeShaders() text, executed on tools/softgl under node ' + process.version,
      layout: 'all arrays zlib+base64; coef_* int16 LE planes; scratch = RGBA8 texture (W/2)x H of pass 1 as stored (row 0 first); ' +
              'planes = RGBA8 textures (W/4) x H = W x H bytes, row 0 = top image row; a P picture predicts from the previous picture of its case',
      cases: out }));
  }
}
main();
