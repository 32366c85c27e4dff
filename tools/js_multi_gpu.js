#!/usr/bin/env node
'use strict';
/*
 * js_multi_gpu.js -- the frame-parallel partition (SURVEY.md 8e) driven from the reference's own host
 * language: N Node processes, one per GPU; process r gets the whole stream and decodes the key-map GOPs
 * g = r (mod N) on device r through the native pipeline (leon_pipeline_config.shard_index / shard_count).
 * Nothing is exchanged between the processes -- closed GOPs share nothing and the key map is the stream's
 * own index (decoders/jsv.js:264-350) -- so there is no collective on this path; the parent adds up rates.
 *   node tools/js_multi_gpu.js <stream.jsv> --gpus N [--loop 64] [--threads 16] [--window 32] [--one-device]
 * --one-device: every shard on device 0 (rehearsal on a one-GPU box).
 */
const path = require('path');
const { spawn } = require('child_process');

const args = process.argv.slice(2);
const file = args.find((a) => !a.startsWith('--'));
const opt = (name, dflt) => { const i = args.indexOf(name); return i >= 0 ? parseInt(args[i + 1], 10) : dflt; };
const n = opt('--gpus', 1), oneDevice = args.includes('--one-device');
const bench = path.join(__dirname, 'js_pipeline_bench.js');
const results = new Array(n);
let left = n, failed = false;
const t0 = process.hrtime.bigint();
for (let r = 0; r < n; r++) {
  const child = spawn(process.execPath, [bench, file, '--loop', String(opt('--loop', 0)), '--threads', String(opt('--threads', 0)),
    '--window', String(opt('--window', 0)), '--device', String(oneDevice ? 0 : r), '--shard-index', String(r), '--shard-count', String(n)].concat(process.argv.includes('--gpu-parser') ? ['--gpu-parser'] : []),
  { stdio: ['ignore', 'pipe', 'inherit'] });
  let out = '';
  child.stdout.on('data', (d) => { out += d; });
  child.on('close', (code) => {
    if (code !== 0) failed = true;
    else results[r] = JSON.parse(out.trim().split('\n').pop());
    if (--left === 0) {
      const wall = Number(process.hrtime.bigint() - t0) / 1e9;
      if (failed) { console.error('a shard failed'); process.exit(1); }
      const pictures = results.reduce((s, x) => s + x.pictures, 0);
      const slowest = Math.max(...results.map((x) => x.seconds));
      console.log(JSON.stringify({ host: 'node ' + process.version, gpus: n, one_device: oneDevice, pictures, seconds_slowest_shard: slowest,
        pictures_per_s: pictures / slowest, wall_seconds_incl_startup: wall,
        per_shard: results.map((x, r) => ({ shard: r, pictures: x.pictures, pictures_per_s: x.pictures_per_s })) }));
    }
  });
}
