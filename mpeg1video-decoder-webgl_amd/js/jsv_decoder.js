'use strict';
/*
 * jsv_decoder.js -- host side of the MI355X-native decode path, in the reference's own
 * language.  Mirrors the decoder object of the reference (window['jsv_dec'],
 * decoders/jsv.js): same method names (_initMeta :237, decodeFrame :426, seek :1618,
 * IDCT_GL :1177), same events ('meta' :262, 'seq' :510, 'frame' :673, 'ended' :437,
 * 'seeked' :1642), same per-picture boundary tensors -- but IDCT_GL() hands them to HIP
 * through the N-API addon (napi/leon_napi.cc -> include/leon.h) instead of issuing WebGL
 * draws, and the bitstream layer is a table-driven parser of its own (vlc_tables.js)
 * that also reads B pictures (ISO/IEC 11172-2; the reference drops them, jsv.js:613-616).
 *
 * What is mirrored on purpose, because it shapes the tensors (SURVEY.md 3.1):
 *   - quantiser-scale and intra maps persist across pictures; skipped macroblocks keep
 *     stale entries (jsv.js:391-393, :794-795)
 *   - coefficient planes, the vector map and the RepAdd map are fresh per non-I picture
 *     (jsv.js:619-649); I pictures overwrite every block
 *   - intra DC is kept in the 0..255 predictor domain, levels are stored raw at their
 *     de-zig-zagged position (jsv.js:1346-1443)
 *   - skipped P macroblocks reset the vector to zero and record it (jsv.js:754-778)
 */
const { EventEmitter } = require('events');
const T = require('./vlc_tables');

const START_PICTURE = 0x00, START_SLICE_FIRST = 0x01, START_SLICE_LAST = 0xAF, START_USER_DATA = 0xB2,
  START_SEQUENCE = 0xC3, START_EXTENSION = 0xB5, START_GOP = 0xB8, START_MAP = 0x000001C4;   // jsv.js:2438-2447
const PICTURE_TYPE_I = 1, PICTURE_TYPE_P = 2, PICTURE_TYPE_B = 3;

class BitReader {
  constructor(bytes) { this.b = bytes; this.pos = 0; this.len = bytes.length * 8; }
  peek(n) {             // n <= 24
    const p = this.pos, i = p >> 3, b = this.b;
    const v = ((b[i] << 24) | ((b[i + 1] | 0) << 16) | ((b[i + 2] | 0) << 8) | (b[i + 3] | 0)) >>> 0;
    return (v << (p & 7)) >>> (32 - n);
  }
  get(n) {              // n <= 32
    if (n > 24) { const hi = this.get(n - 16); return (hi * 65536) + this.get(16); }
    if (n === 0) return 0;
    const v = this.peek(n); this.pos += n; return v;
  }
  skip(n) { this.pos += n; }
  vlc(tab) {
    const e = tab.table[this.peek(tab.maxLen)];
    if (e === 0) throw new Error('invalid VLC at bit ' + this.pos);
    this.pos += e >> 16;
    return e & 0xffff;
  }
  // byte-aligned scan for 00 00 01 xx; returns xx and leaves pos after it, or -1 at the end
  nextStartCode() {
    const b = this.b;
    let i = (this.pos + 7) >> 3;
    for (; i + 3 < b.length; i++) {
      if (b[i] === 0 && b[i + 1] === 0 && b[i + 2] === 1) { this.pos = (i + 4) << 3; return b[i + 3]; }
    }
    this.pos = this.len;
    return -1;
  }
  // the next byte-aligned position holds a start-code prefix (decoders/jsv.js:1710-1760)
  nextBitsAreStartCode() {
    const i = (this.pos + 7) >> 3, b = this.b;
    if (i + 2 >= b.length) return true;
    return b[i] === 0 && b[i + 1] === 0 && b[i + 2] === 1;
  }
}

class JsvDecoder extends EventEmitter {
  /*
   * opts.backend: object exposing the C ABI (see napi/leon_napi.cc): create(cfg) -> handle methods.
   *               null = bitstream layer only (tensor tests without a GPU).
   * opts.nSlots:  output ring size; the reference uses 13 (jsv.js:24).
   */
  constructor(opts) {
    super();
    opts = opts || {};
    this.backendFactory = opts.backend || null;
    this.backend = null;
    this.rendered_frames_n = opts.nSlots || 13;
    this.deviceId = opts.deviceId || 0;
    this.keepTensors = !!opts.keepTensors;
    this.buffer = null;
    this._meta = false;
    this._keyMap = null;
    this._skipTillGop = true;             // player/easybits.player.js:617
    this.sequenceStarted = false;
    this.pictureCodingType = 0;
    this.customIntraQuantMatrix = null;
    this.customNonIntraQuantMatrix = null;
    this._currentTimeSeqUpdate = 0;
    this._ended = false;
    this.slotHolds = new Map();           // slot -> number of holders (decoder references + display)
    this.anchorOld = -1;                  // forward reference of B pictures
    this.anchorNew = -1;                  // latest I/P picture = prev_pic_framebuffer (jsv.js:665)
    this.framesDecoded = 0;
  }

  // ---- input ------------------------------------------------------------------------
  addBuffer(bytes) { this.buffer = new BitReader(bytes instanceof Uint8Array ? bytes : new Uint8Array(bytes)); }

  // ---- container header: decoders/jsv.js:237-313 ----------------------------------------
  _initMeta() {
    const r = this.buffer;
    r.skip(16);
    const meta = { w: r.get(16), h: r.get(16) };
    meta.d = r.get(16) / 100;
    if (!meta.d) { meta.a = r.get(1); meta.d = r.get(23) / 100; }
    // yuva (decoders/jsv.js:256-259: the flag makes the reference allocate four planes per slot, :59-73; it
    // decodes three, :1223).  Here the fourth component is decoded: see decodeMacroblock.
    this.yuva = meta.a === 1;
    this._meta = meta;
    this.emit('meta', meta);
    if (r.peek(24) === 0x000001 && r.b[(r.pos >> 3) + 3] === 0xC4) {
      r.skip(32);
      const count = r.get(32);
      const km = new Uint32Array(count * 2);
      for (let i = 0; i < count; i++) { km[2 * i] = r.get(32); km[2 * i + 1] = r.get(32); }
      this._keyMap = { count, entries: km };
    }
    return true;
  }

  // time of key entry g in seconds: decoders/jsv.js:315-325
  _getTimeByKeyNumber(g) {
    const tc = this._keyMap.entries[2 * g + 1];
    const hour = (tc >>> 26) & 31, minute = (tc >>> 20) & 63, second = (tc >>> 13) & 63, frame = (tc >>> 7) & 63;
    return (hour * 60 + minute) * 60 + second + (frame + 1) / (this.pictureRate || 25);
  }

  // ---- top level: decoders/jsv.js:426-469 ---------------------------------------------
  decodeFrame() {
    if (this._ended) return false;
    const r = this.buffer;
    for (;;) {
      const code = r.nextStartCode();
      if (code < 0) {
        this._flushAnchors();
        this._ended = true;
        this.emit('ended');
        return false;
      }
      if (code === START_SEQUENCE) { this.decodeSequenceHeader(); this._skipTillGop = false; continue; }
      if (this._skipTillGop) continue;
      if (code === START_GOP) { this.decodeGopHeader(); continue; }
      if (code === START_PICTURE && this.sequenceStarted) {
        if (this.decodePicture()) return true;
      }
    }
  }

  // decoders/jsv.js:491-561
  decodeSequenceHeader() {
    const r = this.buffer;
    this.frameWidth = r.get(12);
    this.frameHeight = r.get(12);
    r.skip(4);
    this.pictureRate = T.PICTURE_RATE[r.get(4)];
    r.skip(18 + 1);
    this.bufferSize = 16 * 1024 * r.get(10);
    r.skip(1);
    let intra = null, non = null;
    if (r.get(1)) { intra = new Uint8Array(64); for (let i = 0; i < 64; i++) intra[T.ZIG_ZAG[i]] = r.get(8); }
    if (r.get(1)) { non = new Uint8Array(64); for (let i = 0; i < 64; i++) non[T.ZIG_ZAG[i]] = r.get(8); }
    if (!this.sequenceStarted) this.initBuffers();
    // the stream is honoured for both matrices (decision D7: the reference's GPU path never
    // sees a custom non-intra matrix, jsv.js:556)
    this.intraQuantMatrix = intra || T.DEFAULT_INTRA_QUANT_MATRIX;
    this.nonIntraQuantMatrix = non || T.DEFAULT_NON_INTRA_QUANT_MATRIX;
    if (this.backend) this.backend.setQuantMatrices(this.intraQuantMatrix, this.nonIntraQuantMatrix);
    if (!this.seqSent) {                     // 'send rate only once' (decoders/jsv.js:507-515)
      this.seqSent = true;
      this.emit('seq', { r: this.pictureRate, w: this.frameWidth, h: this.frameHeight });
    }
  }

  // decoders/jsv.js:355-423
  initBuffers() {
    this.mbWidth = (this.frameWidth + 15) >> 4;
    this.mbHeight = (this.frameHeight + 15) >> 4;
    this.mbSize = this.mbWidth * this.mbHeight;
    this.codedWidth = this.mbWidth << 4;
    this.codedHeight = this.mbHeight << 4;
    this.codedSize = this.codedWidth * this.codedHeight;
    this.halfWidth = this.mbWidth << 3;
    this.macroblockQuant = new Uint8Array(this.mbSize);          // persist across pictures
    this.macroblockIsIntra = new Uint8Array(this.mbSize);
    this.currentYDCT16 = new Int16Array(this.codedSize);
    this.currentCbDCT16 = new Int16Array(this.codedSize >> 2);
    this.currentCrDCT16 = new Int16Array(this.codedSize >> 2);
    this.currentADCT16 = this.yuva ? new Int16Array(this.codedSize) : null;
    this.sequenceStarted = true;
    if (this.backendFactory) {
      // = decoder._initGL(gl) + initGLBuffers (player/easybits.player.js:584-585, jsv.js:51-87)
      this.backend = this.backendFactory.create({
        codedWidth: this.codedWidth, codedHeight: this.codedHeight, frameWidth: this.frameWidth,
        frameHeight: this.frameHeight, nSlots: this.rendered_frames_n, deviceId: this.deviceId, alpha: this.yuva ? 1 : 0 });
    }
  }

  // decoders/jsv.js:471-489
  decodeGopHeader() {
    const r = this.buffer;
    r.skip(1);
    const h = r.get(5), m = r.get(6);
    r.skip(1);
    const s = r.get(6), f = r.get(6);
    this._currentTimeSeqUpdate = ((h * 60 + m) * 60 + s + (f + 1) / this.pictureRate) * 1000;
  }

  // decoders/jsv.js:583-676 (+ B pictures)
  decodePicture() {
    const r = this.buffer;
    this.temporalReference = r.get(10);
    const previous = this.pictureCodingType;
    const type = r.get(3);
    r.skip(16);
    if (type <= 0 || type > PICTURE_TYPE_B) return false;
    this.pictureCodingType = type;
    this.macroblockMV = new Int16Array(this.mbSize * 2);
    if (type !== PICTURE_TYPE_I) {
      this.macroblockRepAdd = new Uint8Array(this.mbSize);
      this.fullPelForward = r.get(1);
      this.forwardFCode = r.get(3);
      if (this.forwardFCode === 0) return false;
      this.forwardRSize = this.forwardFCode - 1;
      this.forwardF = 1 << this.forwardRSize;
    }
    if (type === PICTURE_TYPE_B) {
      this.macroblockMVBack = new Int16Array(this.mbSize * 2);
      this.macroblockDir = new Uint8Array(this.mbSize);
      this.fullPelBackward = r.get(1);
      this.backwardFCode = r.get(3);
      if (this.backwardFCode === 0) return false;
      this.backwardRSize = this.backwardFCode - 1;
      this.backwardF = 1 << this.backwardRSize;
    }
    if (type !== PICTURE_TYPE_I) {
      // fresh zeroed coefficient planes (jsv.js:639-649; B pictures need them just the same)
      this.currentYDCT16 = new Int16Array(this.codedSize);
      this.currentCbDCT16 = new Int16Array(this.codedSize >> 2);
      this.currentCrDCT16 = new Int16Array(this.codedSize >> 2);
      if (this.yuva) this.currentADCT16 = new Int16Array(this.codedSize);
    }
    void previous;
    let code;
    do { code = r.nextStartCode(); } while (code === START_EXTENSION || code === START_USER_DATA);
    while (code >= START_SLICE_FIRST && code <= START_SLICE_LAST) {
      this.decodeSlice(code);
      code = r.nextStartCode();
    }
    if (code >= 0) r.pos -= 32;              // rewind(32)
    this.IDCT_GL();
    return true;
  }

  // decoders/jsv.js:683-706
  decodeSlice(slice) {
    const r = this.buffer;
    this.sliceBegin = true;
    this.macroblockAddress = (slice - 1) * this.mbWidth - 1;
    this.motionFwH = this.motionFwHPrev = 0;
    this.motionFwV = this.motionFwVPrev = 0;
    this.motionBwH = this.motionBwHPrev = 0;
    this.motionBwV = this.motionBwVPrev = 0;
    this.prevDir = 0;
    this.dcPredictorY = 128;
    this.dcPredictorCr = 128;
    this.dcPredictorCb = 128;
    this.dcPredictorA = 128;
    this.quantizerScale = r.get(5);
    while (r.get(1)) r.skip(8);
    do { this.decodeMacroblock(); } while (!r.nextBitsAreStartCode());
  }

  // decoders/jsv.js:725-828 (+ B pictures)
  decodeMacroblock() {
    const r = this.buffer, type = this.pictureCodingType, mbw = this.mbWidth;
    let increment = 0, t = r.vlc(T.MBA);
    while (t === 34) t = r.vlc(T.MBA);
    while (t === 35) { increment += 33; t = r.vlc(T.MBA); }
    increment += t;
    if (this.sliceBegin) {
      this.sliceBegin = false;
      this.macroblockAddress += increment;
    } else {
      if (this.macroblockAddress + increment >= this.mbSize) return;
      if (increment > 1) {
        this.dcPredictorY = this.dcPredictorCr = this.dcPredictorCb = this.dcPredictorA = 128;
        if (type === PICTURE_TYPE_P) {
          this.motionFwH = this.motionFwHPrev = 0;
          this.motionFwV = this.motionFwVPrev = 0;
        }
      }
      while (increment > 1) {                  // skipped macroblocks
        const a = ++this.macroblockAddress;
        this.macroblockMV[2 * a] = this.motionFwH;
        this.macroblockMV[2 * a + 1] = this.motionFwV;
        if (type === PICTURE_TYPE_B) {         // same prediction as the previous macroblock
          this.macroblockMVBack[2 * a] = this.motionBwH;
          this.macroblockMVBack[2 * a + 1] = this.motionBwV;
          this.macroblockDir[a] = this.prevDir;
        }
        increment--;
      }
      this.macroblockAddress++;
    }
    const mb = this.macroblockAddress;
    this.mbRow = (mb / mbw) | 0;
    this.mbCol = mb % mbw;
    const mbType = this.macroblockType = r.vlc(T.MBTYPE[type]);
    this.macroblockIntra = mbType & 0x01;
    this.macroblockMotFw = mbType & 0x08;
    this.macroblockMotBw = mbType & 0x04;
    if (mbType & 0x10) this.quantizerScale = r.get(5);
    this.macroblockQuant[mb] = this.quantizerScale;
    this.macroblockIsIntra[mb] = this.macroblockIntra ? 255 : 0;
    if (this.macroblockIntra) {
      this.motionFwH = this.motionFwHPrev = 0;
      this.motionFwV = this.motionFwVPrev = 0;
      this.motionBwH = this.motionBwHPrev = 0;
      this.motionBwV = this.motionBwVPrev = 0;
      this.prevDir = 0;
      if (type !== PICTURE_TYPE_I) this.macroblockRepAdd[mb] = 255;     // jsv.js:1502-1505
    } else {
      this.dcPredictorY = this.dcPredictorCr = this.dcPredictorCb = this.dcPredictorA = 128;
      this.decodeMotionVectors();
      this.macroblockMV[2 * mb] = this.motionFwH;
      this.macroblockMV[2 * mb + 1] = this.motionFwV;
      if (type === PICTURE_TYPE_B) {
        this.macroblockMVBack[2 * mb] = this.motionBwH;
        this.macroblockMVBack[2 * mb + 1] = this.motionBwV;
        this.prevDir = (this.macroblockMotFw ? 1 : 0) | (this.macroblockMotBw ? 2 : 0);
        this.macroblockDir[mb] = this.prevDir;
      }
    }
    const cbp = (mbType & 0x02) ? r.vlc(T.CBP) : (this.macroblockIntra ? 0x3f : 0);
    // yuva (this repo's syntax, tools/jsv_writer.py write_picture -- the reference's slice loop reads six blocks,
    // jsv.js:817-828): four A blocks after Cr, all of them in an intra macroblock, else those named by a 4-bit
    // alpha_pattern that every non-intra macroblock carries at this point
    const apat = this.yuva ? (this.macroblockIntra ? 0xf : r.get(4)) : 0;
    for (let block = 0, mask = 0x20; block < 6; block++, mask >>= 1) {
      if (cbp & mask) this.decodeBlock(block);
    }
    for (let block = 6, mask = 0x8; block < 10; block++, mask >>= 1) {
      if (apat & mask) this.decodeBlock(block);
    }
  }

  _motionComponent(prev, rSize, f) {
    const r = this.buffer;
    let code = r.vlc(T.MOTION), d;
    if (code !== 0 && r.get(1)) code = -code;
    if (code !== 0 && f !== 1) {
      d = ((Math.abs(code) - 1) << rSize) + r.get(rSize) + 1;
      if (code < 0) d = -d;
    } else d = code;
    prev += d;
    if (prev > (f << 4) - 1) prev -= f << 5;
    else if (prev < ((-f) << 4)) prev += f << 5;
    return prev;
  }

  // decoders/jsv.js:831-893 (+ backward vectors)
  decodeMotionVectors() {
    if (this.macroblockMotFw) {
      this.motionFwHPrev = this._motionComponent(this.motionFwHPrev, this.forwardRSize, this.forwardF);
      this.motionFwH = this.fullPelForward ? this.motionFwHPrev << 1 : this.motionFwHPrev;
      this.motionFwVPrev = this._motionComponent(this.motionFwVPrev, this.forwardRSize, this.forwardF);
      this.motionFwV = this.fullPelForward ? this.motionFwVPrev << 1 : this.motionFwVPrev;
    } else if (this.pictureCodingType === PICTURE_TYPE_P) {
      this.motionFwH = this.motionFwHPrev = 0;
      this.motionFwV = this.motionFwVPrev = 0;
    }
    if (this.macroblockMotBw) {
      this.motionBwHPrev = this._motionComponent(this.motionBwHPrev, this.backwardRSize, this.backwardF);
      this.motionBwH = this.fullPelBackward ? this.motionBwHPrev << 1 : this.motionBwHPrev;
      this.motionBwVPrev = this._motionComponent(this.motionBwVPrev, this.backwardRSize, this.backwardF);
      this.motionBwV = this.fullPelBackward ? this.motionBwVPrev << 1 : this.motionBwVPrev;
    }
  }

  // decoders/jsv.js:1338-1525 (decodeBlockGL): raw levels into the component's int16 plane
  decodeBlock(block) {
    const r = this.buffer;
    let plane, stride, base, n = 0;
    if (block < 4 || block >= 6) {                       // luma, or the A component (blocks 6..9, placed like luma)
      const lb = block < 4 ? block : block - 6;
      plane = block < 4 ? this.currentYDCT16 : this.currentADCT16; stride = this.codedWidth;
      base = (this.mbRow * 16 + ((lb >> 1) << 3)) * stride + this.mbCol * 16 + ((lb & 1) << 3);
    } else {
      plane = block === 4 ? this.currentCbDCT16 : this.currentCrDCT16; stride = this.halfWidth;
      base = this.mbRow * 8 * stride + this.mbCol * 8;
    }
    if (this.pictureCodingType === PICTURE_TYPE_I) {     // planes are reused: clear this block first
      for (let y = 0; y < 8; y++) plane.fill(0, base + y * stride, base + y * stride + 8);
    }
    if (this.macroblockIntra) {
      let predictor, size;
      if (block < 4) { predictor = this.dcPredictorY; size = r.vlc(T.DC_LUM); }
      else if (block >= 6) { predictor = this.dcPredictorA; size = r.vlc(T.DC_LUM); }
      else { predictor = block === 4 ? this.dcPredictorCr : this.dcPredictorCb; size = r.vlc(T.DC_CHR); }
      let dc = predictor;
      if (size > 0) {
        const differential = r.get(size);
        dc = (differential & (1 << (size - 1))) ? predictor + differential : predictor + ((-1 << size) | (differential + 1));
      }
      if (block < 4) this.dcPredictorY = dc; else if (block >= 6) this.dcPredictorA = dc;
      else if (block === 4) this.dcPredictorCr = dc; else this.dcPredictorCb = dc;
      plane[base] = dc;
      n = 1;
    }
    for (;;) {
      let run, level;
      const coeff = r.vlc(T.COEF);
      if (coeff === 0x0001 && n > 0 && r.get(1) === 0) break;          // '10' = end of block
      if (coeff === 0xffff) {
        run = r.get(6);
        level = r.get(8);
        if (level === 0) level = r.get(8);
        else if (level === 128) level = r.get(8) - 256;
        else if (level > 128) level -= 256;
      } else {
        run = coeff >> 8;
        level = coeff & 0xff;
        if (r.get(1)) level = -level;
      }
      n += run;
      if (n > 63) throw new Error('coefficient index overflow in macroblock ' + this.macroblockAddress);
      const z = T.ZIG_ZAG[n++];
      plane[base + (z >> 3) * stride + (z & 7)] = level;
    }
  }

  // ---- the drop-in boundary: jsv.prototype.IDCT_GL (decoders/jsv.js:1177-1336) ---------------
  IDCT_GL() {
    const type = this.pictureCodingType;
    const pic = this._boundary(type);
    let ts = 0;
    if (this._currentTimeSeqUpdate) { ts = this._currentTimeSeqUpdate; this._currentTimeSeqUpdate = 0; }
    const frame = { ts, type, temporalReference: this.temporalReference, slot: -1, index: this.framesDecoded++ };
    if (this.keepTensors) frame.tensors = this._snapshot(pic);
    if (this.backend) {
      // = setRenderBuffer (jsv.js:1165-1176): throws "no free render buffers" when the ring is full
      const slot = this.backend.acquireSlot();
      this._hold(slot);                                        // the display's hold
      pic.outSlot = frame.slot = slot;
      if (type === PICTURE_TYPE_P) pic.refFwdSlot = this.anchorNew;
      if (type === PICTURE_TYPE_B) {
        pic.refBwdSlot = this.anchorNew;
        pic.refFwdSlot = this.anchorOld >= 0 ? this.anchorOld : this.anchorNew;   // closed-GOP leading B pictures
      }
      this._submit(pic);
      if (type !== PICTURE_TYPE_B) {                            // prev_pic_framebuffer = framebuffer (jsv.js:665)
        if (this.anchorOld >= 0) this._drop(this.anchorOld);
        this.anchorOld = this.anchorNew;
        this.anchorNew = slot;
        this._hold(slot);                                      // the decoder's hold on its reference
      }
      frame.ybr = [slot, slot, slot];                          // planes of one slot (the reference passes 3 textures)
    }
    this.emit('frame', frame);
  }

  // the per-picture arrays IDCT_GL uploads (decoders/jsv.js:1204-1298)
  _boundary(type) {
    return {
      type,
      coefY: this.currentYDCT16, coefCb: this.currentCbDCT16, coefCr: this.currentCrDCT16, coefA: this.currentADCT16,
      qscale: this.macroblockQuant, intra: this.macroblockIsIntra,
      repadd: type !== PICTURE_TYPE_I ? this.macroblockRepAdd : null,
      mvFwd: type !== PICTURE_TYPE_I ? this.macroblockMV : null,
      mvBwd: type === PICTURE_TYPE_B ? this.macroblockMVBack : null,
      mbDir: type === PICTURE_TYPE_B ? this.macroblockDir : null,
      outSlot: -1, refFwdSlot: -1, refBwdSlot: -1,
    };
  }
  _submit(pic) { this.backend.submitPicture(pic); }

  _snapshot(pic) {
    const c = (a) => (a ? a.slice() : null);
    return { type: pic.type, coefY: c(pic.coefY), coefCb: c(pic.coefCb), coefCr: c(pic.coefCr), coefA: c(pic.coefA), qscale: c(pic.qscale),
      intra: c(pic.intra), repadd: c(pic.repadd), mvFwd: c(pic.mvFwd), mvBwd: c(pic.mvBwd), mbDir: c(pic.mbDir) };
  }

  _hold(slot) { this.slotHolds.set(slot, (this.slotHolds.get(slot) || 0) + 1); }
  _drop(slot) {
    const n = (this.slotHolds.get(slot) || 0) - 1;
    if (n <= 0) { this.slotHolds.delete(slot); if (this.backend) this.backend.releaseSlot(slot); } else this.slotHolds.set(slot, n);
  }
  // = texture.inuse = false in renderFrameGL (player/easybits.player.js:2820), made safe:
  // a slot that is still a reference stays allocated until the decoder lets go of it too
  releaseFrame(frame) { if (frame.slot >= 0) this._drop(frame.slot); }
  _flushAnchors() {
    if (this.anchorOld >= 0) { this._drop(this.anchorOld); this.anchorOld = -1; }
    if (this.anchorNew >= 0) { this._drop(this.anchorNew); this.anchorNew = -1; }
  }

  // = renderFrameGL / YCbCrToRGBA (player/easybits.player.js:2787-2858, :2674-2785)
  renderFrame(frame, flavour) { return this.backend.convertRGBA(frame.slot, flavour | 0); }
  readPlanes(frame) { return this.backend.readPlanes(frame.slot); }

  // decoders/jsv.js:1618-1648: free the ring, jump through the key map, land on a GOP header
  seek(time) {
    if (this.backend) { this._flushAnchors(); this.backend.freeDecodedSlots(); this.slotHolds.clear(); }
    const r = this.buffer;
    let offset = 0;
    if (this._keyMap && this._keyMap.count) {
      let g = Math.min(this._keyMap.count - 1, Math.max(0, Math.floor(this._keyMap.count * time / (this._meta.d || 1))));
      while (g > 0 && this._getTimeByKeyNumber(g) > time + 1e-9) g--;
      while (g + 1 < this._keyMap.count && this._getTimeByKeyNumber(g + 1) <= time + 1e-9) g++;
      offset = this._keyMap.entries[2 * g];
    }
    r.pos = offset * 8;
    this._ended = false;
    this._skipTillGop = true;
    this.emit('seeked', { t: time, offset });
    return offset;
  }

  destroy() { if (this.backend) { this.backend.destroy(); this.backend = null; } }
}

module.exports = { JsvDecoder, BitReader, PICTURE_TYPE_I, PICTURE_TYPE_P, PICTURE_TYPE_B };
