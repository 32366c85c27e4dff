#!/usr/bin/env node
'use strict';
/*
 * cli.js -- command-line front end of the Node host side.
 *   node cli.js tensors <stream.jsv> [--seek=<seconds>]
 *                                             bitstream layer only: boundary tensors per picture
 *                                             (sha256 of each array; small arrays in full) as JSON
 *   node cli.js decode <stream.jsv> [--rgba]  full path through the N-API addon on the GPU:
 *                                             per-frame sha256 of the planes (and RGBA) as JSON
 *   node cli.js seek <stream.jsv> <seconds>   key-map seek, then decode to the end
 *   ... --native                              parse with the native front end (libleon_vlc) and hand
 *                                             pictures over as sparse group lists
 */
const fs = require('fs');
const crypto = require('crypto');
const path = require('path');
const { JsvDecoder } = require('./jsv_decoder');
const { NativeJsvDecoder } = require('./native_decoder');

const sha = (ta) => (ta ? crypto.createHash('sha256').update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest('hex') : null);
const b64 = (ta) => (ta ? Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString('base64') : null);

function loadBackend() {
  // fails loudly when the addon or a GPU is missing: there is no CPU fallback
  return require(path.join(__dirname, '..', 'napi', 'leon_napi.node'));
}

function main() {
  const [cmd, file, ...rest] = process.argv.slice(2);
  if (!cmd || !file) { console.error('usage: cli.js tensors|decode|seek <stream.jsv> ...'); process.exit(2); }
  const bytes = new Uint8Array(fs.readFileSync(file));
  const out = { stream: path.basename(file), events: [], pictures: [] };
  const gpu = cmd !== 'tensors';
  const Decoder = rest.includes('--native') ? NativeJsvDecoder : JsvDecoder;
  const dec = new Decoder({ backend: gpu ? loadBackend() : null, keepTensors: !gpu, nSlots: 13 });
  dec.on('meta', (m) => out.events.push(Object.assign({ ev: 'meta' }, m)));
  dec.on('seq', (s) => out.events.push(Object.assign({ ev: 'seq' }, s)));
  dec.on('seeked', (s) => out.events.push(Object.assign({ ev: 'seeked' }, s)));
  dec.on('ended', () => out.events.push({ ev: 'ended' }));
  dec.on('frame', (f) => {
    out.events.push({ ev: 'frame', ts: f.ts });
    const p = { type: f.type, ts: f.ts, temporalReference: f.temporalReference };
    if (!gpu) {
      const t = f.tensors;
      p.sha = { coefY: sha(t.coefY), coefCb: sha(t.coefCb), coefCr: sha(t.coefCr), coefA: sha(t.coefA), qscale: sha(t.qscale), intra: sha(t.intra),
        repadd: sha(t.repadd), mvFwd: sha(t.mvFwd), mvBwd: sha(t.mvBwd), mbDir: sha(t.mbDir) };
      p.qscale = b64(t.qscale); p.intra = b64(t.intra); p.repadd = b64(t.repadd);
      p.mvFwd = b64(t.mvFwd); p.mvBwd = b64(t.mvBwd); p.mbDir = b64(t.mbDir);
    } else {
      const planes = dec.readPlanes(f);
      p.slot = f.slot;
      p.planes = { y: sha(planes.y), cb: sha(planes.cb), cr: sha(planes.cr), a: sha(planes.a) };
      if (rest.includes('--rgba')) p.rgba = sha(dec.renderFrame(f, 0));
      dec.releaseFrame(f);                 // = renderFrameGL clearing texture.inuse
    }
    out.pictures.push(p);
  });
  dec.addBuffer(bytes);
  dec._initMeta();
  if (cmd === 'seek') dec.seek(parseFloat(rest.filter((a) => !a.startsWith('--'))[0]));
  const seekOpt = rest.find((a) => a.startsWith('--seek='));
  if (seekOpt) dec.seek(parseFloat(seekOpt.slice(7)));
  while (dec.decodeFrame());
  out.codedWidth = dec.codedWidth; out.codedHeight = dec.codedHeight; out.mbWidth = dec.mbWidth; out.mbHeight = dec.mbHeight;
  dec.destroy();
  process.stdout.write(JSON.stringify(out));
}
main();
