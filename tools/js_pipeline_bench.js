#!/usr/bin/env node
'use strict';
/*
 * js_pipeline_bench.js -- the native pipeline driven from Node: the JavaScript thread starts it and then
 * only receives 'frames' events (napi_threadsafe_function); an interval timer counts how often the event
 * loop got to run meanwhile (the reference's page would be frozen inside decodeFrame for the duration).
 *   node tools/js_pipeline_bench.js <stream.jsv> [--loop 64] [--threads 16] [--window 32] [--hash] [--gl]
 *        [--device d --shard-index r --shard-count N]   (one process per GPU: tools/js_multi_gpu.js)
 * --hash: print the sha256 of every frame instead (tests; small streams).
 * --gl: displayFlavour 1 -- the frames in the fp32 arithmetic of the page's renderFrameGL (include/leon_pipeline.h).
 */
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const { LeonPipeline } = require(path.join(__dirname, '..', 'mpeg1video-decoder-webgl_amd', 'js', 'leon_pipeline.js'));

const args = process.argv.slice(2);
const file = args.find((a) => !a.startsWith('--'));
const opt = (name, dflt) => { const i = args.indexOf(name); return i >= 0 ? parseInt(args[i + 1], 10) : dflt; };
const hash = args.includes('--hash');
const stream = fs.readFileSync(file);
let ticks = 0, frames = 0, windows = 0;
const out = [];
const timer = setInterval(() => { ticks++; }, 1);
const t0 = process.hrtime.bigint();
const p = new LeonPipeline(stream, { parserThreads: opt('--threads', 0), gopsPerWindow: opt('--window', 0),
                                     windowsInFlight: opt('--inflight', 0), loop: opt('--loop', 0), autoRelease: !hash,
                                     deviceId: opt('--device', 0), shardIndex: opt('--shard-index', 0), shardCount: opt('--shard-count', 0),
                                     gpuParser: process.argv.includes('--gpu-parser') ? 1 : -1, displayFlavour: args.includes('--gl') ? 1 : 0 });
p.on('frames', (w, fs_) => {
  windows++;
  frames += fs_.length;
  if (hash) {
    for (const f of fs_) out.push({ gop: f.gop, displayIndex: f.displayIndex, type: f.type, ts: f.ts,
                                    sha256: crypto.createHash('sha256').update(p.readFrame(w, f.index)).digest('hex') });
    p.releaseWindow(w);
  }
});
p.on('error', (e) => { console.error(String(e)); process.exitCode = 1; });
p.on('ended', () => {
  clearInterval(timer);
  const dt = Number(process.hrtime.bigint() - t0) / 1e9;
  const s = p.stats();
  console.log(JSON.stringify({ host: 'node ' + process.version, path: 'native pipeline (leon_pipeline_*), frames by napi_threadsafe_function',
    pictures: frames, windows, seconds: s.seconds, wall_seconds: dt, pictures_per_s: frames / s.seconds,
    event_loop_ticks_while_decoding: ticks, width: s.frameWidth, height: s.frameHeight, parserThreads: s.parserThreads,
    gopsPerWindow: s.gopsPerWindow, frames: hash ? out : undefined }));
  p.destroy();
});
