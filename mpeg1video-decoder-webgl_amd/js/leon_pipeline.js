'use strict';
/*
 * leon_pipeline.js -- the native decode pipeline (include/leon_pipeline.h) for a JavaScript host.
 *
 * The reference's page pulls pictures one at a time on its only thread: decodeFrame() -> bit-serial
 * parse -> IDCT_GL, frame by frame, paced by the display timer (decoders/jsv.js:426-469,
 * player/easybits.player.js:2310-2324, :2543-2617).  Here the whole loop runs in native threads (GOP-shard
 * parsers, one submit thread, one notify thread) and the host only receives events, through a
 * napi_threadsafe_function -- the JavaScript thread never waits for the parser or the GPU.
 *
 *   const { LeonPipeline } = require('./leon_pipeline');
 *   const p = new LeonPipeline(fs.readFileSync('clip.jsv'), { parserThreads: 16, gopsPerWindow: 32 });
 *   p.on('frame', (f) => ...)      // {gop, displayIndex, type, ts, window, index}: same event name as the
 *                                  // reference decoder's (decoders/jsv.js:673); frames arrive in display order
 *   p.on('frames', (window, frames) => ...)   // one call per window; keep it with {autoRelease: false}
 *   p.on('ended', () => ...)       // decoders/jsv.js:437
 *   p.on('error', (err) => ...)
 *   p.readFrame(window, index) -> Uint8Array RGBA (copies to the host: tests, thumbnails)
 *   p.releaseWindow(window); p.stats(); p.destroy();
 */
const path = require('path');
const EventEmitter = require('events');

class LeonPipeline extends EventEmitter {
  constructor(stream, opts) {
    super();
    opts = opts || {};
    const addon = opts.backend || require(path.join(__dirname, '..', 'napi', 'leon_napi.node'));
    if (!Buffer.isBuffer(stream)) stream = Buffer.from(stream.buffer, stream.byteOffset, stream.byteLength);
    this.autoRelease = opts.autoRelease !== false;
    this.ended = false;
    this._p = addon.createPipeline(stream, opts, (w, frames, status) => this._deliver(w, frames, status));
  }

  _deliver(window, frames, status) {
    if (window < 0) {
      this.ended = true;
      if (status) this.emit('error', new Error('leon pipeline stopped with status ' + status));
      this.emit('ended');
      return;
    }
    if (status) {
      this.emit('error', new Error('leon pipeline: window ' + window + ' failed with status ' + status));
      this._p.releaseWindow(window);
      return;
    }
    frames.forEach((f, i) => { f.window = window; f.index = i; });
    this.emit('frames', window, frames);
    for (const f of frames) this.emit('frame', f);
    if (this.autoRelease) this._p.releaseWindow(window);
  }

  // a stream that is still arriving (opts.validBytes at construction; the Buffer has the file's final size): the loader
  // writes the next chunk into the same Buffer and reports how far it is valid now -- features/bitreader.js:332 addBuffer
  feed(validBytes) { this._p.feed(validBytes); }
  readFrame(window, index) { return this._p.readFrame(window, index); }
  releaseWindow(window) { this._p.releaseWindow(window); }
  stats() { return this._p.stats(); }
  destroy() { if (this._p) { this._p.destroy(); this._p = null; } }
}

module.exports = { LeonPipeline };
