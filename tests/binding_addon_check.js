#!/usr/bin/env node
/*
 * binding_addon_check.js -- the SHIPPED binding (mpeg1video-decoder-webgl_amd/js/reference_binding.js) driving the REAL
 * addon (napi/leon_napi.node -> libleon_hip.so -> HIP) on the GPU box, where /root/reference does not exist.
 *
 * What stands in for the reference's decoder object: a plain object that carries the reference's FIELD NAMES -- the ones the
 * binding reads in initGLBuffers / setRenderBuffer / IDCT_GL (decoders/jsv.js:51-87, :1165-1176, :1177-1336):
 *   codedWidth, codedHeight, frameWidth, frameHeight, rendered_frames_n (13, jsv.js:24), n_comps,
 *   pictureCodingType, currentYDCT16 / currentCbDCT16 / currentCrDCT16, macroblockQuant, macroblockIsIntra,
 *   macroblockRepAdd, macroblockMV, framebuffer, prev_pic_framebuffer, glFrameBuffers
 * -- filled, picture by picture, from tests/golden/glsl_idct_cases.json: the tensors that were handed to the unmodified
 * reference's IDCT_GL on tools/softgl when the fixture was made, and the planes its shaders produced.  The driver below does
 * what decodePicture does around the call (jsv.js:662-673): IDCT_GL(), prev_pic_framebuffer = framebuffer, a 'frame' whose
 * textures the display releases with `texture.inuse = false` (player/easybits.player.js:2820).  Nothing of the reference
 * travels: field names and data only.  (tests/test_reference_binding.py holds the same binding against the reference's own
 * decodeFrame loop with a recording stand-in, in the build container.)
 *
 *   node tests/binding_addon_check.js tests/golden/glsl_idct_cases.json      -> one JSON object on stdout
 */
'use strict';
const fs = require('fs');
const path = require('path');
const zlib = require('zlib');
const ROOT = path.join(__dirname, '..');
const binding = require(path.join(ROOT, 'mpeg1video-decoder-webgl_amd', 'js', 'reference_binding.js'));
const addon = require(path.join(ROOT, 'mpeg1video-decoder-webgl_amd', 'napi', 'leon_napi.node'));

const unz = (s) => { const b = zlib.inflateSync(Buffer.from(s, 'base64')); return new Uint8Array(b.buffer, b.byteOffset, b.length).slice(); };
const i16 = (u8) => new Int16Array(u8.buffer, u8.byteOffset, u8.length >> 1);
const same = (a, b) => a.length === b.length && Buffer.compare(Buffer.from(a.buffer, a.byteOffset, a.length), Buffer.from(b.buffer, b.byteOffset, b.length)) === 0;

// the stand-in for jsv_dec: its prototype gets the binding's methods, an instance the reference's fields
function StubDecoder(cw, ch) {
  this.codedWidth = cw; this.codedHeight = ch;
  this.frameWidth = cw; this.frameHeight = ch;
  this.mbWidth = cw >> 4; this.mbHeight = ch >> 4;
  this.rendered_frames_n = 13;              // decoders/jsv.js:24
  this.n_comps = 3;
  this.framebuffer = null;
  this.prev_pic_framebuffer = null;
}
binding.apply(StubDecoder.prototype, addon);

function load(dec, p) {                      // what the slice loop leaves behind for IDCT_GL (jsv.js:1204-1298)
  dec.pictureCodingType = p.type;
  dec.currentYDCT16 = i16(unz(p.coef_y));
  dec.currentCbDCT16 = i16(unz(p.coef_cb));
  dec.currentCrDCT16 = i16(unz(p.coef_cr));
  dec.macroblockQuant = unz(p.qscale);
  dec.macroblockIsIntra = unz(p.intra);
  if (p.type === 2) {
    dec.macroblockRepAdd = unz(p.repadd);
    dec.macroblockMV = i16(unz(p.mv_fwd));
  }
}

function runCase(c) {
  const dec = new StubDecoder(c.coded_w, c.coded_h);
  // QUANT_MATRIX reaches the binding through the texImage2D of its inert GL object (jsv.js:139-145): the same entry point
  dec._leonSetMatrices(unz(c.quant_matrices), true);
  dec.initGLBuffers();
  const out = { name: c.name, pictures: [], slots: [] };
  const shown = [];
  for (const p of c.pictures) {
    load(dec, p);
    dec.IDCT_GL();
    dec.prev_pic_framebuffer = dec.framebuffer;                                      // jsv.js:665
    const frame = { ybr: [0, 1, 2].map((k) => dec.prev_pic_framebuffer[k].texture) };   // jsv.js:673
    dec.leon.sync();
    const slot = frame.ybr[0].slot;
    const got = dec.leon.readPlanes(slot);
    const want = p.planes.map(unz);
    out.pictures.push({ type: p.type, slot, inuse: frame.ybr[0].inuse, y: same(got.y, want[0]), cb: same(got.cb, want[1]), cr: same(got.cr, want[2]),
                        oneTexturePerSlot: frame.ybr[0] === frame.ybr[1] && frame.ybr[1] === frame.ybr[2] });
    out.slots.push(slot);
    shown.push(frame);
    // the display is done with the frame BEFORE the next picture has predicted from it (player.js:2820 runs on the
    // page's animation frame): the binding holds the release back until the P picture has been submitted
    frame.ybr[0].inuse = false;
  }
  // the released slots come back: the ring never grows past two live slots in this loop
  out.distinctSlots = Array.from(new Set(out.slots)).length;
  // ring exhaustion: nobody releases -> the 14th picture throws what the reference throws (jsv.js:1175)
  const hog = new StubDecoder(c.coded_w, c.coded_h);
  hog._leonSetMatrices(unz(c.quant_matrices), true);
  hog.initGLBuffers();
  load(hog, c.pictures[0]);
  let thrown = null, taken = 0;
  try {
    for (let i = 0; i < 14; i++) { hog.IDCT_GL(); hog.prev_pic_framebuffer = hog.framebuffer; taken++; }
  } catch (e) { thrown = String(e.message || e); }
  out.exhaustion = { taken, thrown };
  // a seek frees every slot (jsv.js:1618-1648 -> GLfreeDecodedBuffers) and decoding goes on
  hog.GLfreeDecodedBuffers();
  let after = null;
  try { hog.IDCT_GL(); after = hog.framebuffer[0].texture.slot; } catch (e) { after = String(e.message || e); }
  out.afterFreeDecodedBuffers = after;
  // the display conversion for a host without a canvas, GL flavour = the reference's live display path
  const rgba = dec.leonConvertRGBA({ ybr: shown[shown.length - 1].ybr }, 1);
  out.rgbaBytes = rgba ? rgba.length : 0;
  hog.leon.destroy();
  dec.leon.destroy();
  return out;
}

const fixture = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
process.stdout.write(JSON.stringify({ abi: addon.abiVersion(), cases: fixture.cases.map(runCase) }));
