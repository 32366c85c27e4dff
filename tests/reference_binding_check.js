#!/usr/bin/env node
/*
 * reference_binding_check.js -- applies mpeg1video-decoder-webgl_amd/js/reference_binding.js to the UNMODIFIED
 * reference decoder (read from /root/reference when the test runs; build container only -- nothing of the reference
 * is stored here) with a RECORDING stand-in for the N-API addon, decodes a stream through the reference's own
 * decodeFrame loop and prints what the binding handed to the library: one record per call, arrays as SHA-256.
 *
 *   node tests/reference_binding_check.js /root/reference tests/golden/streams/tiny_ip_32x32.jsv [--real]
 *
 * --real: the stand-in forwards every call to the real addon (GPU box) and adds the planes of every frame.
 * tests/test_reference_binding.py compares the records with tests/golden/parser_*.json (the tensors the same reference
 * uploaded through WebGL, recorded by tools/make_golden.js).
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const crypto = require('crypto');
const binding = require(path.join(__dirname, '..', 'mpeg1video-decoder-webgl_amd', 'js', 'reference_binding.js'));

const REF = process.argv[2];
const STREAM = process.argv[3];
const REAL = process.argv.includes('--real');
const HONOUR = process.argv.includes('--honour-non-intra');

const sha = (a) => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');

function loadReference() {
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
  for (const f of ['features/eventdispatcher.js', 'features/bitreader.js', 'decoders/shaders/mpeg1video.js', 'decoders/jsv.js'])
    vm.runInContext(fs.readFileSync(path.join(REF, f), 'utf8'), sandbox, { filename: f });
  return sandbox;
}

// the addon's surface (napi/leon_napi.cc), recording; slots behave like leon_acquire_slot / leon_release_slot
function recordingAddon(calls, real) {
  return {
    create(cfg) {
      calls.push({ call: 'create', cfg });
      const inner = real ? real.create(cfg) : null;
      const inuse = new Array(cfg.nSlots).fill(false);
      return {
        setQuantMatrices(intra, nonIntra) {
          calls.push({ call: 'setQuantMatrices', intra: intra ? sha(intra) : null, nonIntra: nonIntra ? sha(nonIntra) : null });
          if (inner) inner.setQuantMatrices(intra, nonIntra);
        },
        acquireSlot() {
          const s = inuse.indexOf(false);
          if (s < 0) throw new Error('no free render buffers');
          inuse[s] = true;
          if (inner && inner.acquireSlot() !== s) throw new Error('the library picked another slot');
          calls.push({ call: 'acquireSlot', slot: s });
          return s;
        },
        releaseSlot(s) { if (!inuse[s]) throw new Error('release of a free slot ' + s); inuse[s] = false; calls.push({ call: 'releaseSlot', slot: s }); if (inner) inner.releaseSlot(s); },
        freeDecodedSlots() { inuse.fill(false); calls.push({ call: 'freeDecodedSlots' }); if (inner) inner.freeDecodedSlots(); },
        submitPicture(p) {
          const rec = { call: 'submitPicture', type: p.type, outSlot: p.outSlot, refFwdSlot: p.refFwdSlot, keys: Object.keys(p).sort() };
          for (const k of ['coefY', 'coefCb', 'coefCr', 'qscale', 'intra', 'repadd', 'mvFwd'])
            if (p[k] !== undefined) rec[k] = { ctor: p[k].constructor.name, length: p[k].length, sha256: sha(p[k]) };
          calls.push(rec);
          if (inner) inner.submitPicture(p);
        },
        convertRGBA(slot, flavour) { return inner ? inner.convertRGBA(slot, flavour) : null; },
        readPlanes(slot) { return inner ? inner.readPlanes(slot) : null; },
        sync() { if (inner) inner.sync(); },
        destroy() { if (inner) inner.destroy(); },
      };
    },
  };
}

function main() {
  const ctx = loadReference();
  const calls = [];
  const real = REAL ? require(path.join(__dirname, '..', 'mpeg1video-decoder-webgl_amd', 'napi', 'leon_napi.node')) : null;
  binding.apply(ctx.jsv_dec.prototype, recordingAddon(calls, real), { honourNonIntraMatrix: HONOUR });
  const dec = new ctx.jsv_dec();
  dec._initGL({ thisObjectIsIgnored: true });                 // player/easybits.player.js:585 hands a WebGL context in
  const events = [], frames = [];
  dec.on('meta', (e) => events.push({ ev: 'meta', w: e.detail.w, h: e.detail.h }));
  dec.on('seq', (e) => events.push({ ev: 'seq', r: e.detail.r, w: e.detail.w, h: e.detail.h }));
  dec.on('ended', () => events.push({ ev: 'ended' }));
  dec.on('frame', (e) => {
    const f = e.detail;
    const rec = { ts: f.ts, slot: f.ybr[0].slot, sameTexture: f.ybr[0] === f.ybr[1] && f.ybr[1] === f.ybr[2] };
    if (REAL) {
      dec.leon.sync();
      const p = dec.leon.readPlanes(rec.slot);
      rec.planes = { y: sha(p.y), cb: sha(p.cb), cr: sha(p.cr) };
    }
    frames.push(rec);
    calls.push({ call: 'frame', slot: rec.slot });
    events.push({ ev: 'frame', ts: f.ts });
    // the display releases the frame (player/easybits.player.js:2820) -- but the reference's P pictures still predict
    // from it (SURVEY.md section 5: the latent reuse hazard).  Like tools/make_golden.js, release at once.
    f.ybr[0].inuse = 0;
  });
  const bytes = new Uint8Array(fs.readFileSync(STREAM));
  dec.buffer.addBuffer({ data: bytes, start: 0, end: bytes.length - 1, total: bytes.length });
  dec._initMeta();
  dec._skipTillGop = true;
  const ended = () => events.length > 0 && events[events.length - 1].ev === 'ended';
  for (let i = 0; i < 100000 && !ended(); i++) {
    const before = events.length;
    dec.decodeFrame();
    if (events.length === before && i > 50000) break;
  }
  // a seek frees every slot (decoders/jsv.js:1618-1648 -> GLfreeDecodedBuffers)
  if (process.argv.includes('--seek')) { dec.setRenderBuffer(); dec.GLfreeDecodedBuffers(); }
  process.stdout.write(JSON.stringify({ mbWidth: dec.mbWidth, mbHeight: dec.mbHeight, codedWidth: dec.codedWidth, codedHeight: dec.codedHeight,
                                        nSlots: dec.rendered_frames_n, events, frames, calls }));
}
main();
