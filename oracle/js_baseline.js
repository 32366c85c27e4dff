#!/usr/bin/env node
'use strict';
/*
 * js_baseline.js -- runner of the plain-JavaScript oracle (oracle/leon_oracle.js): SURVEY.md 8d's CPU
 * baseline "in the reference's own language", timed with process.hrtime on one thread and on
 * worker_threads.  Test / baseline infrastructure only.
 *
 *   node js_baseline.js check <dir>                      sha256 of every picture's planes and RGBA (JSON)
 *   node js_baseline.js time <dir> <seconds> <threads>   whole GOPs (decode + RGBA) until the budget is
 *                                                        used: one thread, then <threads> workers (JSON)
 * <dir> holds manifest.json and raw little-endian arrays written by oracle/oracle_py.py:dump_gop().
 */
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const O = require('./leon_oracle.js');

function load(dir) {
  const m = JSON.parse(fs.readFileSync(path.join(dir, 'manifest.json'), 'utf8'));
  const arr = (name, T) => {
    if (!name) return null;
    const b = fs.readFileSync(path.join(dir, name));
    return new T(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength));
  };
  m.qm = Uint8Array.from(m.qm);
  for (const p of m.gop) {
    p.coefY = arr(p.files.coef_y, Int16Array); p.coefCb = arr(p.files.coef_cb, Int16Array); p.coefCr = arr(p.files.coef_cr, Int16Array);
    p.qscale = arr(p.files.qscale, Uint8Array); p.intra = arr(p.files.intra, Uint8Array);
    p.repadd = arr(p.files.repadd, Uint8Array); p.mbDir = arr(p.files.mb_dir, Uint8Array);
    p.mvFwd = arr(p.files.mv_fwd, Int16Array); p.mvBwd = arr(p.files.mv_bwd, Int16Array);
  }
  return m;
}

// one GOP in coded order: decode + RGBA of every picture; returns the planes by display index
function decodeGop(m, withRgba, sink) {
  const n = m.cw * m.ch * 3 / 2, outs = {};
  for (const p of m.gop) {
    const out = new Uint8Array(n);
    const fwd = p.fwd !== null ? p.fwd : p.bwd;
    O.decodePicture(p, m.cw, m.ch, m.qm, fwd !== null ? outs[fwd] : null, p.bwd !== null ? outs[p.bwd] : null, out);
    outs[p.disp] = out;
    if (withRgba) { const rgba = O.ycbcrToRgba(out, m.cw, m.ch, m.fw, m.fh); if (sink) sink(p, out, rgba); }
  }
  return outs;
}

function timeLoop(m, seconds) {
  const t0 = process.hrtime.bigint();
  let pictures = 0, dt = 0;
  do {
    decodeGop(m, true, null);
    pictures += m.gop.length;
    dt = Number(process.hrtime.bigint() - t0) / 1e9;
  } while (dt < seconds);
  return { pictures, seconds: dt };
}

const wt = require('worker_threads');
if (!wt.isMainThread) {
  const m = load(wt.workerData.dir);
  wt.parentPort.postMessage(timeLoop(m, wt.workerData.seconds));
} else {
  const [cmd, dir, secArg, thrArg] = process.argv.slice(2);
  if (cmd === 'check') {
    const m = load(dir), res = [];
    const sha = (a) => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');
    decodeGop(m, true, (p, planes, rgba) => res.push({ disp: p.disp, type: p.type, planes: sha(planes), rgba: sha(rgba) }));
    console.log(JSON.stringify(res));
  } else if (cmd === 'time') {
    const seconds = parseFloat(secArg || '5'), threads = parseInt(thrArg || '0', 10) || require('os').cpus().length;
    const m = load(dir);
    const one = timeLoop(m, seconds);
    const mbs = (m.cw >> 4) * (m.ch >> 4);
    const t0 = process.hrtime.bigint();
    let done = 0, pictures = 0;
    for (let i = 0; i < threads; i++) {
      const w = new wt.Worker(__filename, { workerData: { dir, seconds } });
      w.on('message', (r) => {
        pictures += r.pictures;
        if (++done === threads) {
          const dt = Number(process.hrtime.bigint() - t0) / 1e9;
          console.log(JSON.stringify({ node: process.version, macroblocks_per_picture: mbs,
            one_thread: { pictures: one.pictures, seconds: one.seconds, macroblocks_per_s: one.pictures * mbs / one.seconds },
            workers: { threads, pictures, seconds: dt, macroblocks_per_s: pictures * mbs / dt } }));
        }
      });
      w.on('error', (e) => { console.error(String(e)); process.exit(1); });
    }
  } else {
    console.error('usage: js_baseline.js check|time <dir> [seconds] [threads]');
    process.exit(2);
  }
}
