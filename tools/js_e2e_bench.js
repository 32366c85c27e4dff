#!/usr/bin/env node
'use strict';
/*
 * js_e2e_bench.js -- the drop-in path as a JavaScript host drives it, on one Node thread:
 * stream bytes -> decoder object (JavaScript parser + dense boundary, or native front end + sparse
 * boundary with --native) -> N-API addon -> libleon_hip -> planes in device memory.
 *   node tools/js_e2e_bench.js <stream.jsv> [--native] [--seconds 8] [--threads 0]
 */
const fs = require('fs');
const path = require('path');
const root = path.join(__dirname, '..', 'mpeg1video-decoder-webgl_amd');
const { JsvDecoder } = require(path.join(root, 'js', 'jsv_decoder.js'));
const { NativeJsvDecoder } = require(path.join(root, 'js', 'native_decoder.js'));
const backend = require(path.join(root, 'napi', 'leon_napi.node'));

const args = process.argv.slice(2);
const file = args.find((a) => !a.startsWith('--'));
const native = args.includes('--native');
const opt = (name, dflt) => { const i = args.indexOf(name); return i >= 0 ? parseFloat(args[i + 1]) : dflt; };
const seconds = opt('--seconds', 8), threads = opt('--threads', 0);
const bytes = new Uint8Array(fs.readFileSync(file));
const Decoder = native ? NativeJsvDecoder : JsvDecoder;
const dec = new Decoder({ backend, nSlots: 13, threads });
let n = 0;
dec.on('frame', (f) => { n++; dec.releaseFrame(f); });      // displayed at once: the slot goes back to the ring
dec.addBuffer(bytes);
dec._initMeta();
const t0 = process.hrtime.bigint();
let dt = 0;
do {
  while (dec.decodeFrame());
  dec.seek(0);
  dt = Number(process.hrtime.bigint() - t0) / 1e9;
} while (dt < seconds);
dec.backend.sync();
dt = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({ host: 'node ' + process.version, parser: native ? 'libleon_vlc (native, sparse boundary)' : 'JavaScript (dense boundary)',
  pictures: n, seconds: dt, pictures_per_s: n / dt, width: dec.frameWidth, height: dec.frameHeight }));
dec.destroy();
