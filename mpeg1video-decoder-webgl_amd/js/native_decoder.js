'use strict';
/*
 * native_decoder.js -- JsvDecoder with the bitstream layer moved into native code
 * (SURVEY.md 8f #1).  Same object surface as jsv_decoder.js / the reference decoder
 * (decoders/jsv.js: _initMeta :237, decodeFrame :426, seek :1618, IDCT_GL :1177; events 'meta',
 * 'seq', 'frame', 'ended', 'seeked'), but
 *   - the stream is parsed by libleon_vlc.so through napi/leon_vlc_napi.node: table-driven codes,
 *     the slices of a picture on worker threads (the reference: one bit at a time on the page's
 *     only thread, jsv.js:1593-1599), and
 *   - IDCT_GL() hands the picture over as SPARSE group lists (include/leon_vlc.h) via
 *     backend.submitSparse -- about 4 bytes per non-zero coefficient instead of the 6.27 MB of
 *     dense int16 planes a 1080p picture uploads in the reference (jsv.js:1237-1243).
 * The reconstruction, the slot ring and the anchor bookkeeping are inherited unchanged.
 */
const path = require('path');
const { JsvDecoder, PICTURE_TYPE_I, PICTURE_TYPE_B } = require('./jsv_decoder');

function loadVlc() {
  return require(path.join(__dirname, '..', 'napi', 'leon_vlc_napi.node'));
}

// sparse group lists -> the dense planes of the reference boundary (tests, tensor snapshots)
function densify(info, pic) {
  const cw = info.codedWidth, ch = info.codedHeight, hw = cw >> 1;
  const y = new Int16Array(cw * ch), cb = new Int16Array((cw * ch) >> 2), cr = new Int16Array((cw * ch) >> 2);
  const nY = 2 * info.mbHeight * info.groupsY, nC = info.mbHeight * info.groupsC;
  const a = info.nGroups > nY + 2 * nC ? new Int16Array(cw * ch) : null;       // yuva: the A groups follow Cr's
  for (let g = 0; g < info.nGroups; g++) {
    let plane, stride, R, gg;
    if (g < nY) { plane = y; stride = cw; R = (g / info.groupsY) | 0; gg = g % info.groupsY; }
    else if (g >= nY + 2 * nC) { const k = g - nY - 2 * nC; plane = a; stride = cw; R = (k / info.groupsY) | 0; gg = k % info.groupsY; }
    else { const k = (g - nY) % nC; plane = g - nY < nC ? cb : cr; stride = hw; R = (k / info.groupsC) | 0; gg = k % info.groupsC; }
    for (let e = pic.grpOff[g]; e < pic.grpOff[g + 1]; e++) {
      const v = pic.entries[e], off = (v >>> 16) & 1023;
      const r = off >> 7, b = (off >> 4) & 7, c = (off >> 1) & 7;
      plane[(R * 8 + r) * stride + (gg * 8 + b) * 8 + c] = (v << 16) >> 16;
    }
  }
  return { coefY: y, coefCb: cb, coefCr: cr, coefA: a };
}

class NativeJsvDecoder extends JsvDecoder {
  /* opts as JsvDecoder, plus opts.threads (0 = one per hardware thread, at most 16) */
  constructor(opts) {
    super(opts);
    this.threads = (opts && opts.threads) | 0;
    this.vlc = null;
    this.native = null;
    this.info = null;
  }

  addBuffer(bytes) {
    if (this.native) this.native.close();
    this.vlc = this.vlc || loadVlc();
    this.native = this.vlc.open(bytes instanceof Uint8Array ? bytes : new Uint8Array(bytes), this.threads);
    this.info = this.native.info();
  }

  _initMeta() {
    const i = this.info;
    this._meta = { w: i.frameWidth, h: i.frameHeight, d: i.duration };
    if (i.hasAlpha >= 0) this._meta.a = i.hasAlpha;
    this._keyMap = i.keymapCount ? { count: i.keymapCount } : null;
    this.emit('meta', this._meta);
    return true;
  }

  _sequence(i) {
    this.frameWidth = i.frameWidth; this.frameHeight = i.frameHeight; this.pictureRate = i.pictureRate;
    this.intraQuantMatrix = i.intraQm; this.nonIntraQuantMatrix = i.nonIntraQm;
    if (!this.sequenceStarted) {
      this.mbWidth = i.mbWidth; this.mbHeight = i.mbHeight; this.mbSize = i.mbWidth * i.mbHeight;
      this.codedWidth = i.codedWidth; this.codedHeight = i.codedHeight; this.codedSize = i.codedWidth * i.codedHeight;
      this.halfWidth = i.mbWidth << 3;
      this.sequenceStarted = true;
      if (this.backendFactory) {
        this.backend = this.backendFactory.create({
          codedWidth: this.codedWidth, codedHeight: this.codedHeight, frameWidth: this.frameWidth,
          frameHeight: this.frameHeight, nSlots: this.rendered_frames_n, deviceId: this.deviceId,
          alpha: this._meta && this._meta.a === 1 ? 1 : 0 });       // yuva: the sparse lists carry the A groups
      }
    }
    if (this.backend) this.backend.setQuantMatrices(this.intraQuantMatrix, this.nonIntraQuantMatrix);
    if (!this.seqSent) { this.seqSent = true; this.emit('seq', { r: this.pictureRate, w: this.frameWidth, h: this.frameHeight }); }
  }

  // = decodeFrame (decoders/jsv.js:426-469): one picture per call
  decodeFrame() {
    if (this._ended) return false;
    const p = this.native.nextPicture();
    if (!p) {
      this._flushAnchors();
      this._ended = true;
      this.emit('ended');
      return false;
    }
    if (p.newSequence || !this.sequenceStarted) { this.info = this.native.info(); this._sequence(this.info); }
    this.pictureCodingType = p.type;
    this.temporalReference = p.temporalReference;
    if (p.ts) this._currentTimeSeqUpdate = p.ts;
    this.macroblockQuant = p.qscale; this.macroblockIsIntra = p.intra;
    this.macroblockRepAdd = p.repadd; this.macroblockMV = p.mvFwd;
    this.macroblockMVBack = p.mvBwd; this.macroblockDir = p.mbDir;
    this._sparse = p;
    this.IDCT_GL();
    return true;
  }

  // the sparse twin of the boundary object IDCT_GL() builds
  _boundary(type) {
    const p = this._sparse;
    return {
      type, grpOff: p.grpOff, entries: p.entries, nEntries: p.nEntries,
      qscale: p.qscale, intra: p.intra,
      repadd: type !== PICTURE_TYPE_I ? p.repadd : null, mvFwd: type !== PICTURE_TYPE_I ? p.mvFwd : null,
      mvBwd: type === PICTURE_TYPE_B ? p.mvBwd : null, mbDir: type === PICTURE_TYPE_B ? p.mbDir : null,
      outSlot: -1, refFwdSlot: -1, refBwdSlot: -1,
    };
  }
  _submit(pic) { this.backend.submitSparse(pic); }
  _snapshot(pic) {
    const d = densify(this.info, pic);
    const c = (a) => (a ? a.slice() : null);
    return { type: pic.type, coefY: d.coefY, coefCb: d.coefCb, coefCr: d.coefCr, coefA: d.coefA, qscale: c(pic.qscale), intra: c(pic.intra),
      repadd: c(pic.repadd), mvFwd: c(pic.mvFwd), mvBwd: c(pic.mvBwd), mbDir: c(pic.mbDir),
      nEntries: pic.nEntries };
  }

  seek(time) {
    if (this.backend) { this._flushAnchors(); this.backend.freeDecodedSlots(); this.slotHolds.clear(); }
    const offset = this.native.seek(time);
    this._ended = false;
    this.emit('seeked', { t: time, offset });
    return offset;
  }

  destroy() {
    if (this.native) { this.native.close(); this.native = null; }
    super.destroy();
  }
}

module.exports = { NativeJsvDecoder, densify };
