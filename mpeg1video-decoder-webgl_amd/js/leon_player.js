'use strict';
/*
 * leon_player.js -- headless, HTML5-<video>-shaped façade over JsvDecoder (SURVEY.md 8b, 8f #3).
 *
 * The reference wraps its decoder in a <video>-like element (window['video_jsv'],
 * player/easybits.player.js:157-315: getters/setters :160-203, play :2235, pause, load,
 * currentTime= -> seek :1423-1481, frame queue of at most MAX_DECODED_FRAMES = 10
 * player/parts/end.js:57, display timer displayFrame :2451-2505, 'frame' handler onf :2543-2668).
 * This class keeps that surface where it makes sense without a DOM: the same property and
 * event names, the same ready/network state constants (end.js:39-52), the same queue bound, and
 * the same hand-off -- decode ahead into the slot ring, display one frame per frame duration,
 * release the slot when it has been rendered (= uint8Y.inuse = false, player.js:2820).
 * Network loading, audio sync, controls, poster and bitrate switching are out of scope.
 *
 * Beyond the reference: B pictures arrive in coded order and are re-ordered for display
 * (an I/P picture is shown when the next I/P picture has been decoded).
 *
 * opts.pipeline = true puts the native pipeline (leon_pipeline.js, include/leon_pipeline.h) under the same surface:
 * parsing, reconstruction and display conversion run in native threads, whole GOPs arrive as 'frames' events in
 * display order, the JavaScript thread only paces and hands frames out.  The queue bound is then the pipeline's:
 * one GOP per window, two windows in flight.  Seeking recreates the pipeline at the key-map entry for the time.
 */
const fs = require('fs');
const { EventEmitter } = require('events');
const { JsvDecoder, PICTURE_TYPE_B } = require('./jsv_decoder');
const { NativeJsvDecoder } = require('./native_decoder');

const NETWORK_EMPTY = 0, NETWORK_IDLE = 1, NETWORK_LOADING = 2, NETWORK_NO_SOURCE = 3;
const HAVE_NOTHING = 0, HAVE_METADATA = 1, HAVE_CURRENT_DATA = 2, HAVE_FUTURE_DATA = 3, HAVE_ENOUGH_DATA = 4;
const MAX_DECODED_FRAMES = 10;               // player/parts/end.js:57

class LeonPlayer extends EventEmitter {
  /*
   * opts.backend   the N-API addon module (required to decode pixels; null = bitstream only)
   * opts.realtime  true: one frame per frame duration on a timer (default); false: as fast as possible
   * opts.render    (rgba: Uint8Array, frame) => void, called for every displayed frame when a backend is present
   * opts.nativeParser  true: parse with libleon_vlc on worker threads and hand pictures over as sparse
   *                    group lists (native_decoder.js) instead of the JavaScript bitstream layer
   * opts.pipeline      true: the native pipeline does everything below this class (needs frame width % 8 == 0);
   *                    opts.parserThreads (default 4)
   */
  constructor(opts) {
    super();
    this.opts = Object.assign({ realtime: true, backend: null, render: null, flavour: 0, nativeParser: false, pipeline: false, parserThreads: 4 }, opts || {});
    this._pipe = null;
    this._windowLeft = new Map();      // pipeline mode: window id -> frames not displayed yet
    this.networkState = NETWORK_EMPTY;
    this.readyState = HAVE_NOTHING;
    this.paused = true;
    this.ended = false;
    this.seeking = false;
    this.loop = false;
    this.autoplay = false;
    this.playbackRate = 1;
    this.error = null;
    this.duration = NaN;
    this.videoWidth = 0;
    this.videoHeight = 0;
    this._src = '';
    this._currentTime = 0;
    this._decoder = null;
    this._decodedFrames = [];          // display-ordered, bounded by MAX_DECODED_FRAMES
    this._heldAnchor = null;           // I/P picture waiting for its display turn
    this._timer = null;
    this._streamEnded = false;
    this.framesDisplayed = 0;
  }

  canPlayType(type) { return /jsv/i.test(type || '') ? 'probably' : ''; }

  get src() { return this._src; }
  set src(v) { this._src = v; this.load(); }
  get currentSrc() { return typeof this._src === 'string' ? this._src : '[buffer]'; }

  get currentTime() { return this._currentTime; }
  set currentTime(t) { this._seek(t); }

  // ---- load: player/easybits.player.js:496-658 without the network ------------------------
  load() {
    this._stopTimer();
    this._stopPipeline();
    if (this._decoder) this._decoder.destroy();
    this.networkState = NETWORK_LOADING;
    this.readyState = HAVE_NOTHING;
    this.ended = false;
    this._streamEnded = false;
    this._decodedFrames = [];
    this._heldAnchor = null;
    this._currentTime = 0;
    this.emit('loadstart');
    let bytes;
    try {
      bytes = typeof this._src === 'string' ? new Uint8Array(fs.readFileSync(this._src)) : new Uint8Array(this._src);
    } catch (e) {
      this.networkState = NETWORK_NO_SOURCE;
      this.error = { code: 4, message: String(e.message) };       // MEDIA_ERR_SRC_NOT_SUPPORTED (end.js:20-26)
      this.emit('error', this.error);
      return;
    }
    if (this.opts.pipeline) { this._bytes = bytes; this._startPipeline(0); if (this.autoplay) this.play(); return; }
    const Decoder = this.opts.nativeParser ? NativeJsvDecoder : JsvDecoder;
    const d = this._decoder = new Decoder({ backend: this.opts.backend, nSlots: 13 });
    d.on('meta', (m) => { this.duration = m.d; });
    d.on('seq', (s) => {
      this.videoWidth = s.w; this.videoHeight = s.h; this.frameDuration = 1000 / s.r;
      if (this.readyState < HAVE_METADATA) { this.readyState = HAVE_METADATA; this.emit('loadedmetadata'); }
    });
    d.on('frame', (f) => this._onFrame(f));
    d.on('ended', () => { this._streamEnded = true; this._flushAnchor(); });
    d.on('seeked', () => { this.seeking = false; this.emit('seeked'); });
    d.addBuffer(bytes);
    d._initMeta();
    this.networkState = NETWORK_IDLE;
    this._fill();                                  // decode ahead: first frame -> loadeddata / canplay
    if (this.autoplay) this.play();
  }

  // ---- pipeline mode: the native pipeline in place of the decoder object ---------------------------
  _startPipeline(startSeconds) {
    const { LeonPipeline } = require('./leon_pipeline');
    this._stopPipeline();
    const p = this._pipe = new LeonPipeline(Buffer.from(this._bytes.buffer, this._bytes.byteOffset, this._bytes.byteLength),
      { backend: this.opts.backend, parserThreads: this.opts.parserThreads, gopsPerWindow: 1, windowsInFlight: 2, autoRelease: false, startSeconds,
        gpuParser: this.opts.gpuParser === undefined ? 0 : (this.opts.gpuParser ? 1 : -1),       // opts.gpuParser: the slice layer on the GPU (include/leon_pipeline.h)
        displayFlavour: this.opts.displayFlavour | 0 });      // 1: the fp32 arithmetic of the page's renderFrameGL (player/parts/end.js:77-156)
    const s = p.stats();
    this.videoWidth = s.frameWidth; this.videoHeight = s.frameHeight; this.frameDuration = 1000 / (s.pictureRate || 25);
    if (s.duration) this.duration = s.duration;
    if (this.readyState < HAVE_METADATA) { this.readyState = HAVE_METADATA; this.emit('loadedmetadata'); }
    this.networkState = NETWORK_IDLE;
    p.on('frames', (w, frames) => {
      if (p !== this._pipe) return;                 // a pipeline that has been seeked away from
      this._windowLeft.set(w, frames.length);
      for (const f of frames) this._decodedFrames.push(f);      // display order already
      if (this.readyState < HAVE_CURRENT_DATA) {
        this.readyState = HAVE_CURRENT_DATA; this.emit('loadeddata');
        this.readyState = HAVE_FUTURE_DATA; this.emit('canplay');
      }
      if (this._decodedFrames.length >= MAX_DECODED_FRAMES && this.readyState < HAVE_ENOUGH_DATA) { this.readyState = HAVE_ENOUGH_DATA; this.emit('canplaythrough'); }
      if (this.seeking) { this.seeking = false; this.emit('seeked'); }
      if (!this.paused && !this.opts.realtime) this._drain();
    });
    p.on('ended', () => { if (p === this._pipe) { this._streamEnded = true; if (!this.paused && !this.opts.realtime) this._drain(); } });
    p.on('error', (e) => { this.error = { code: 3, message: String(e.message) }; this.emit('error', this.error); });     // MEDIA_ERR_DECODE
  }
  _stopPipeline() {
    if (!this._pipe) return;
    const p = this._pipe;
    this._pipe = null;
    this._windowLeft.clear();
    p.destroy();
  }
  _drain() { while (!this.paused && !this.ended && (this._decodedFrames.length || this._streamEnded)) this._displayFrame(); }

  // ---- 'frame' handler: onf, player/easybits.player.js:2543-2668 ----------------------------
  _onFrame(f) {
    if (f.type === PICTURE_TYPE_B) {
      this._decodedFrames.push(f);                 // B pictures are displayed as they come
    } else {
      this._flushAnchor();                         // the previous anchor's turn has come
      this._heldAnchor = f;
    }
    if (this.readyState < HAVE_CURRENT_DATA && this._decodedFrames.length) {
      this.readyState = HAVE_CURRENT_DATA;
      this.emit('loadeddata');
      this.readyState = HAVE_FUTURE_DATA;
      this.emit('canplay');
    }
  }
  _flushAnchor() { if (this._heldAnchor) { this._decodedFrames.push(this._heldAnchor); this._heldAnchor = null; } }

  // keep decoding while the queue is short (player.js:2613-2617); the slot ring (13) bounds it too
  _fill() {
    if (this._pipe || this.opts.pipeline) return;   // the pipeline decodes ahead by itself
    const d = this._decoder;
    while (!this._streamEnded && this._decodedFrames.length < MAX_DECODED_FRAMES) {
      if (!d.decodeFrame()) break;
    }
    if (this._decodedFrames.length >= MAX_DECODED_FRAMES && this.readyState < HAVE_ENOUGH_DATA) {
      this.readyState = HAVE_ENOUGH_DATA;
      this.emit('canplaythrough');
    }
  }

  // ---- play / pause: player/easybits.player.js:2235-2308 -----------------------------------
  play() {
    if (!this._decoder && !this._pipe) this.load();
    if (this.ended) { this._seek(0); this.ended = false; }
    if (!this.paused) return;
    this.paused = false;
    this.emit('play');
    this.emit('playing');
    if (this.opts.realtime) this._timer = setInterval(() => this._displayFrame(), (this.frameDuration || 40) / this.playbackRate);
    else if (this._pipe) this._drain();            // frames arrive as events: displayed as they come
    else while (!this.paused && !this.ended) this._displayFrame();
  }
  pause() {
    if (this.paused) return;
    this.paused = true;
    this._stopTimer();
    this.emit('pause');
  }
  _stopTimer() { if (this._timer) { clearInterval(this._timer); this._timer = null; } }

  // ---- displayFrame: player/easybits.player.js:2451-2505 -----------------------------------
  _displayFrame() {
    if (!this._decodedFrames.length) this._fill();
    const f = this._decodedFrames.shift();
    if (!f) {
      if (this._streamEnded) {
        if (this.loop) { this._seek(0); return; }
        this.ended = true;
        this.paused = true;
        this._stopTimer();
        this.emit('ended');
      } else {
        this.emit('waiting');
      }
      return;
    }
    if (f.ts) this._currentTime = f.ts / 1000; else this._currentTime += (this.frameDuration || 40) / 1000;
    if (this._pipe) {
      // = renderFrameGL: the frame is RGBA in device memory already; a renderer gets a host copy
      if (this.opts.render) this.opts.render(this._pipe.readFrame(f.window, f.index), f);
      const left = this._windowLeft.get(f.window) - 1;
      if (left > 0) this._windowLeft.set(f.window, left);
      else { this._windowLeft.delete(f.window); this._pipe.releaseWindow(f.window); }     // = texture.inuse = false for the GOP
    } else if (this._decoder.backend) {
      // = renderFrameGL (player.js:2787-2858), then the slot goes back to the ring (:2820)
      if (this.opts.render) this.opts.render(this._decoder.renderFrame(f, this.opts.flavour), f);
      this._decoder.releaseFrame(f);
    }
    this.framesDisplayed++;
    this.emit('timeupdate', { currentTime: this._currentTime, frame: f });
    this._fill();                                  // ask for more (player.js:2504)
  }

  // ---- currentTime= -> sct -> decoder.seek: player/easybits.player.js:1423-1481, jsv.js:1618 ----
  _seek(t) {
    if (this.opts.pipeline && this._bytes) {
      this.seeking = true;
      this.emit('seeking');
      this._decodedFrames = [];
      this._streamEnded = false;
      this.ended = false;
      this._currentTime = t;
      this._startPipeline(t);                       // 'seeked' when its first frames arrive
      return;
    }
    if (!this._decoder) return;
    this.seeking = true;
    this.emit('seeking');
    for (const f of this._decodedFrames) this._decoder.releaseFrame(f);
    if (this._heldAnchor) this._decoder.releaseFrame(this._heldAnchor);
    this._decodedFrames = [];
    this._heldAnchor = null;
    this._streamEnded = false;
    this.ended = false;
    this._decoder.seek(t);
    this._currentTime = t;
    this._fill();
  }

  destroy() { this._stopTimer(); this._stopPipeline(); if (this._decoder) { this._decoder.destroy(); this._decoder = null; } }
}

Object.assign(LeonPlayer, { NETWORK_EMPTY, NETWORK_IDLE, NETWORK_LOADING, NETWORK_NO_SOURCE,
  HAVE_NOTHING, HAVE_METADATA, HAVE_CURRENT_DATA, HAVE_FUTURE_DATA, HAVE_ENOUGH_DATA, MAX_DECODED_FRAMES });
module.exports = { LeonPlayer };
