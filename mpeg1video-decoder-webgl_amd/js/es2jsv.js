#!/usr/bin/env node
'use strict';
/*
 * es2jsv.js -- MPEG-1 video elementary stream <-> JSV container (SURVEY.md 8f #4).
 *
 * A JSV file is an MPEG-1 video elementary stream whose sequence start code is 0xC3 instead of
 * 0xB3 (decoders/jsv.js:2438-2447), behind a small header the reference reads in _initMeta
 * (decoders/jsv.js:237-313): magic (16 bits), frame width and height (16 each), duration in 1/100 s
 * (16 bits, or 0 followed by an alpha flag and 23 bits), then optionally START_MAP 00 00 01 C4, an
 * entry count and (byte offset of a sequence header, GOP time code) pairs -- the random-access index
 * seek() jumps through (:1618-1648).
 *
 *   node es2jsv.js to-jsv <in.m1v> <out.jsv>     node es2jsv.js to-es <in.jsv> <out.m1v>
 */
const PICTURE_RATE = [0, 23.976, 24, 25, 29.97, 30, 50, 59.94, 60, 15, 5, 10, 12, 15, 0, 0];   // decoders/jsv.js:1762-1765

function* startCodes(b, from) {
  for (let i = from || 0; i + 3 < b.length; i++) {
    if (b[i] === 0 && b[i + 1] === 0 && b[i + 2] === 1) { yield [i, b[i + 3]]; i += 3; }
  }
}

function esToJsv(es, opts) {
  opts = opts || {};
  let first = -1, w = 0, h = 0, rate = 25, pictures = 0, pendingSeq = -1;
  const keys = [];
  for (const [i, code] of startCodes(es, 0)) {
    if (code === 0xB3) {
      if (first < 0) {
        first = i;
        w = (es[i + 4] << 4) | (es[i + 5] >> 4);
        h = ((es[i + 5] & 15) << 8) | es[i + 6];
        rate = PICTURE_RATE[es[i + 7] & 15] || 25;
      }
      pendingSeq = i;
    } else if (code === 0xB8 && pendingSeq >= 0) {
      // the 25 time-code bits of the GOP header, left-aligned: exactly the key map's layout
      const tc = (((es[i + 4] << 24) | (es[i + 5] << 16) | (es[i + 6] << 8) | es[i + 7]) >>> 7) << 7;
      keys.push([pendingSeq - first, tc >>> 0]);
      pendingSeq = -1;
    } else if (code === 0x00 && first >= 0) {
      pictures++;
      pendingSeq = -1;
    }
  }
  if (first < 0) throw new Error('no sequence header (00 00 01 B3) in the stream');
  const body = Uint8Array.from(es.subarray(first));
  for (const [i, code] of startCodes(body, 0)) if (code === 0xB3) body[i + 3] = 0xC3;
  const map = opts.keyMap === false ? [] : keys;
  const nHdr = 11 + (map.length ? 8 + 8 * map.length : 0);
  const out = new Uint8Array(nHdr + body.length);
  const dv = new DataView(out.buffer);
  dv.setUint16(0, 0x6A73); dv.setUint16(2, w); dv.setUint16(4, h); dv.setUint16(6, 0);
  const d = Math.round(pictures / rate * 100) & 0x7fffff;       // alpha flag 0 + 23 bits
  out[8] = (d >> 16) & 0x7f; out[9] = (d >> 8) & 255; out[10] = d & 255;
  if (map.length) {
    out.set([0, 0, 1, 0xC4], 11);
    dv.setUint32(15, map.length);
    map.forEach(([off, tc], k) => { dv.setUint32(19 + 8 * k, off + nHdr); dv.setUint32(23 + 8 * k, tc); });
  }
  out.set(body, nHdr);
  return out;
}

function jsvToEs(jsv) {
  let first = -1;
  for (const [i, code] of startCodes(jsv, 0)) if (code === 0xC3) { first = i; break; }
  if (first < 0) throw new Error('no sequence header (00 00 01 C3) in the stream');
  const es = Uint8Array.from(jsv.subarray(first));
  for (const [i, code] of startCodes(es, 0)) if (code === 0xC3) es[i + 3] = 0xB3;
  return es;
}

module.exports = { esToJsv, jsvToEs };

if (require.main === module) {
  const fs = require('fs');
  const [cmd, src, dst] = process.argv.slice(2);
  if (!src || !dst || (cmd !== 'to-jsv' && cmd !== 'to-es')) { console.error('usage: es2jsv.js to-jsv|to-es <in> <out>'); process.exit(2); }
  const inp = new Uint8Array(fs.readFileSync(src));
  fs.writeFileSync(dst, Buffer.from(cmd === 'to-jsv' ? esToJsv(inp) : jsvToEs(inp)));
}
