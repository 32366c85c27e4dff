'use strict';
/*
 * reference_binding.js -- the patch that puts the HIP path behind the REFERENCE's own decoder object.
 *
 *   const binding = require('./reference_binding.js');
 *   binding.apply(window.jsv_dec.prototype, require('../napi/leon_napi.node'));     // once, after decoders/jsv.js is loaded
 *   // player/easybits.player.js:584-585 stays as it is:  decoder = new window['jsv_dec'];  decoder._initGL(gl);
 *
 * What it replaces (paths under the reference):
 *   _initGL          decoders/jsv.js:88-208   runs UNCHANGED, but against an inert GL object of this file instead of the
 *                                             WebGL context it is handed: programs, shaders and textures become plain
 *                                             objects, nothing is compiled.  The one upload that matters outside the hot
 *                                             path -- QUANT_MATRIX to TextureDCTQuant, at init (:139-145) and on a sequence
 *                                             header with custom matrices (:540-558) -- becomes leon_set_quant_matrices.
 *   initGLBuffers    decoders/jsv.js:51-87    leon_create; the 13-entry ring `glFrameBuffers.rendered` keeps its shape
 *                                             ([slot][component] = {framebuffer, texture}) because decodePicture (:665-673)
 *                                             and the player (player/easybits.player.js:2787-2858) reach into it
 *   setRenderBuffer  decoders/jsv.js:1165     leon_acquire_slot (same policy: the first slot not in use)
 *   IDCT_GL          decoders/jsv.js:1177     leon_submit_picture with the decoder's own arrays -- the hot-path entry
 *   GLfreeDecodedBuffers  decoders/jsv.js:1160    leon_free_decoded_slots
 * A displayed frame is released the way the reference does it, `_frame.ybr[0].inuse = false`
 * (player/easybits.player.js:2820): `inuse` is an accessor on the slot's texture objects that calls leon_release_slot.
 *
 * The slice loop, the bit reader, the events and the player are the reference's, untouched.  Nothing here decodes,
 * dequantises or predicts anything: arrays in, slot numbers out.
 *
 * `addon` is mpeg1video-decoder-webgl_amd/napi/leon_napi.node ({create(cfg) -> decoder}); tests hand in a recording
 * stand-in with the same surface (tests/reference_binding_check.js).
 *
 * opts.honourNonIntraMatrix (default false): the reference binds the never-created `TextureDCTQuantInter` for a custom
 * non-intra matrix (decoders/jsv.js:556) -- WebGL drops that upload, and its shaders keep the default matrix.  The
 * default mirrors that; true hands the stream's non-intra matrix to the kernels as this repository's own decoder does.
 */

// the GL constants _initGL, createTexture and decodeSequenceHeader read (WebGL 1 values)
const GL_CONST = {
  TEXTURE_2D: 0x0DE1, LUMINANCE: 0x1909, LUMINANCE_ALPHA: 0x190A, RGBA: 0x1908, UNSIGNED_BYTE: 0x1401, FLOAT: 0x1406,
  NEAREST: 0x2600, CLAMP_TO_EDGE: 0x812F, TEXTURE_MAG_FILTER: 0x2800, TEXTURE_MIN_FILTER: 0x2801, TEXTURE_WRAP_S: 0x2802,
  TEXTURE_WRAP_T: 0x2803, UNPACK_ALIGNMENT: 0x0CF5, FRAMEBUFFER: 0x8D40, COLOR_ATTACHMENT0: 0x8CE0,
  VERTEX_SHADER: 0x8B31, FRAGMENT_SHADER: 0x8B30, COMPILE_STATUS: 0x8B81, LINK_STATUS: 0x8B82, TRIANGLE_STRIP: 0x0005,
  HIGH_FLOAT: 0x8DF2, MEDIUM_FLOAT: 0x8DF1, MEDIUM_INT: 0x8DF4, HIGH_INT: 0x8DF5, ARRAY_BUFFER: 0x8892, STATIC_DRAW: 0x88E4,
};
for (let i = 0; i < 16; i++) GL_CONST['TEXTURE' + i] = 0x84C0 + i;

// An object that answers the GL calls of the reference's set-up code and does nothing -- except remember which texture
// is bound, so that a QUANT_MATRIX upload can be told from the others.
function inertGL(decoder, opts) {
  let bound = null, serial = 0;
  const gl = Object.assign({}, GL_CONST, {
    getShaderPrecisionFormat: () => ({ rangeMin: 31, rangeMax: 30, precision: 0 }),      // "integer" flavour (jsv.js:98-103)
    getShaderParameter: () => true,
    getProgramParameter: () => true,
    getShaderInfoLog: () => '',
    getProgramInfoLog: () => '',
    createTexture: () => ({ leon: 'texture', id: ++serial }),
    createProgram: () => ({ leon: 'program', id: ++serial }),
    createShader: () => ({ leon: 'shader', id: ++serial }),
    createFramebuffer: () => ({ leon: 'framebuffer', id: ++serial }),
    createBuffer: () => ({ leon: 'buffer', id: ++serial }),
    getUniformLocation: (_p, name) => ({ uniform: name }),
    getAttribLocation: () => 0,
    bindTexture(_target, tex) { bound = tex; },
    texImage2D(_target, _level, _ifmt, w, h, _border, fmt, _type, data) {
      // QUANT_MATRIX: rows 0-7 the intra matrix, rows 8-15 the non-intra one (jsv.js:139-145, :545-558)
      if (w !== 8 || h !== 16 || fmt !== GL_CONST.LUMINANCE || !data || data.length < 128) return;
      const onQuantTexture = bound !== null && bound !== undefined && bound === decoder.TextureDCTQuant;
      if (onQuantTexture || opts.honourNonIntraMatrix) decoder._leonSetMatrices(data, onQuantTexture);
    },
  });
  // everything else (pixelStorei, attachShader, linkProgram, uniform1i, vertexAttribPointer, ...) is a no-op
  return new Proxy(gl, { get: (t, name) => (name in t ? t[name] : () => undefined) });
}

function apply(proto, addon, opts) {
  opts = opts || {};
  const referenceInitGL = proto._initGL;

  proto._initGL = function (_webglContext) {
    // TextureDCTQuant is assigned from createTexture() and uploaded to in the next statement (jsv.js:144-145):
    // bindTexture has seen the object by then, the comparison in texImage2D needs the field -- which is set
    return referenceInitGL.call(this, inertGL(this, opts));
  };

  // the matrices the kernels use: what sits in TextureDCTQuant.  Until leon_create has run they wait here.
  proto._leonSetMatrices = function (quant128, wholeTexture) {
    const m = this._leonMatrices || (this._leonMatrices = { intra: null, nonIntra: null });
    if (wholeTexture) {
      m.intra = Uint8Array.from(quant128.subarray(0, 64));
      m.nonIntra = Uint8Array.from(quant128.subarray(64, 128));
    } else {
      m.nonIntra = Uint8Array.from(quant128.subarray(64, 128));       // honourNonIntraMatrix only
    }
    if (this.leon) this.leon.setQuantMatrices(m.intra, m.nonIntra);
  };

  proto.initGLBuffers = function () {                                  // decoders/jsv.js:51-87
    this.leon = addon.create({ codedWidth: this.codedWidth, codedHeight: this.codedHeight, frameWidth: this.frameWidth,
                               frameHeight: this.frameHeight, nSlots: this.rendered_frames_n, deviceId: opts.deviceId | 0 });
    if (this._leonMatrices) this.leon.setQuantMatrices(this._leonMatrices.intra, this._leonMatrices.nonIntra);
    const release = (slot) => this.leon.releaseSlot(slot);
    const rendered = new Array(this.rendered_frames_n);
    for (let i = 0; i < this.rendered_frames_n; i++) {
      // one texture object per slot, shared by its components: the reference marks and releases a slot through
      // component 0 only (jsv.js:1162, :1170; player.js:2820)
      const texture = slotTexture(i, release);
      rendered[i] = new Array(this.n_comps);
      for (let comp = 0; comp < this.n_comps; comp++) rendered[i][comp] = { framebuffer: { slot: i, comp }, texture };
    }
    this.glFrameBuffers = { rendered, idct_1d: [] };                   // the pass-1 scratch lives inside the kernel (LDS)
  };

  proto.setRenderBuffer = function () {                                // decoders/jsv.js:1165-1176
    const slot = this.leon.acquireSlot();                              // throws "no free render buffers" (LEON_ERR_NO_FREE_SLOT)
    const fbo = this.glFrameBuffers.rendered[slot];
    fbo[0].texture._taken();
    this.framebuffer = fbo;
  };

  proto.GLfreeDecodedBuffers = function () {                           // decoders/jsv.js:1160-1164 (seek, :1623)
    for (let j = 0; j < this.rendered_frames_n; j++) this.glFrameBuffers.rendered[j][0].texture._dropped();
    this._leonReference = null;
    this.leon.freeDecodedSlots();                                      // all of them at once
  };

  proto.IDCT_GL = function () {                                        // decoders/jsv.js:1177-1336, called at :662
    this.setRenderBuffer();                                            // :1301
    const predicted = this.pictureCodingType === 2;                    // PICTURE_TYPE_P
    const pic = {
      type: this.pictureCodingType,
      outSlot: this.framebuffer[0].texture.slot,
      refFwdSlot: predicted ? this.prev_pic_framebuffer[0].texture.slot : -1,      // :1320, set at :665
      coefY: this.currentYDCT16, coefCb: this.currentCbDCT16, coefCr: this.currentCrDCT16,   // :1179-1183, :1243
      qscale: this.macroblockQuant,                                    // :1206
      intra: this.macroblockIsIntra,                                   // :1217
    };
    if (predicted) {
      pic.repadd = this.macroblockRepAdd;                              // :1284
      pic.mvFwd = this.macroblockMV;                                   // :1298 (uploaded through its byte view)
    }
    this.leon.submitPicture(pic);
    // this picture is the next one's forward reference (:665); the previous reference is done with
    const before = this._leonReference;
    this._leonReference = this.framebuffer[0].texture;
    this._leonReference._pin();
    if (before && before !== this._leonReference) before._unpin();
  };

  // renderFrameGL's counterpart for a host without a canvas: the frame event's textures carry the slot
  proto.leonConvertRGBA = function (frame, flavour) { return this.leon.convertRGBA(frame.ybr[0].slot, flavour | 0); };
  return proto;
}

// The object the reference knows as a slot's texture: `uid`, and `inuse` -- which the player clears when the frame has
// been displayed (player/easybits.player.js:2820).  Clearing it releases the slot in the library -- unless the slot is
// still the forward reference of the next picture (prev_pic_framebuffer, decoders/jsv.js:665): the reference would hand
// such a slot out again and render into the texture it is sampling (SURVEY.md section 5); here the release waits until
// the picture that predicts from it has been submitted (launches on one decoder are ordered).
function slotTexture(slot, release) {
  let inuse = 0, pinned = false, owed = false;
  const let_go = () => { if (pinned) owed = true; else release(slot); };
  return Object.defineProperties({ uid: slot, slot }, {
    inuse: { enumerable: true, get: () => inuse, set(v) { const was = inuse; inuse = v ? 1 : 0; if (was && !inuse) let_go(); } },
    _taken: { value() { inuse = 1; owed = false; } },                   // setRenderBuffer: the library has marked it already
    _dropped: { value() { inuse = 0; pinned = owed = false; } },       // GLfreeDecodedBuffers: the library frees every slot in one call
    _pin: { value() { pinned = true; } },
    _unpin: { value() { pinned = false; if (owed) { owed = false; release(slot); } } },
  });
}

module.exports = { apply, inertGL, GL_CONST };
