'use strict';
/*
 * leon_oracle.js -- CPU ORACLE in plain JavaScript (test / baseline infrastructure, NOT the product).
 *
 * The same restatement as oracle/leon_oracle.c, in the reference's own language and shaped like the
 * CPU code the reference carries: per-macroblock prediction in the manner of copyMacroblock
 * (decoders/jsv.js:895-1129), a per-block two-pass integer IDCT following the shader text
 * (decoders/shaders/mpeg1video.js:19-29, composed as decoders/jsv.js:2461-2464; decisions D1-D10 of
 * SURVEY.md 8c) and the RGB conversion of YCbCrToRGBA (player/easybits.player.js:2674-2785).
 * It exists for SURVEY.md 8d's CPU baseline -- "the build's own plain-JS single-thread restatement",
 * timed on one thread and on worker_threads (oracle/js_baseline.js) -- and is pinned to the C oracle
 * bit for bit by tests/test_oracle_js.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 */

const PREMULTIPLIER = new Uint8Array([                    // decoders/jsv.js:1797-1806
  32, 44, 42, 38, 32, 25, 17, 9, 44, 62, 58, 52, 44, 35, 24, 12,
  42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
  32, 44, 42, 38, 32, 25, 17, 9, 25, 35, 33, 30, 25, 20, 14, 7,
  17, 24, 23, 20, 17, 14, 9, 5, 9, 12, 12, 10, 9, 7, 5, 2]);
const Y04 = Math.fround(0.4);                             // _y, binary32

// mpeg1video.js:23 / :26; '/' = int division truncating toward zero (D1).  X, o: Int32Array(8)
function butterfly8(X, o) {
  const b1 = X[4], b3 = X[2] + X[6], b4 = X[5] - X[3], t1 = X[1] + X[7], t2 = X[3] + X[5];
  const b6 = X[1] - X[7], b7 = t1 + t2, m0 = X[0];
  const x4 = (((b6 * 473 - b4 * 196 + 128) / 256) | 0) - b7;
  const x0 = x4 - ((((t1 - t2) * 362 + 128) / 256) | 0);
  const x1 = m0 - b1;
  const x2 = ((((X[2] - X[6]) * 362 + 128) / 256) | 0) - b3;
  const x3 = m0 + b1;
  const y3 = x1 + x2, y4 = x3 + b3, y5 = x1 - x2, y6 = x3 - b3;
  const y7 = -x0 - (((b4 * 473 + b6 * 196 + 128) / 256) | 0);
  o[0] = b7 + y4; o[1] = x4 + y3; o[2] = y5 - x0; o[3] = y6 - y7;
  o[4] = y6 + y7; o[5] = x0 + y5; o[6] = y3 - x4; o[7] = y4 - b7;
}

// _B(), the RGBA8 store and _E() (mpeg1video.js:18): int16 wrap, high byte saturating
function handoffStore(w) {
  const v = w < 0 ? w + 65536 : w;
  let hi = Math.floor(v / 256);
  const lo = v - hi * 256;
  if (hi < 0) hi = 0;
  if (hi > 255) hi = 255;
  const u = hi * 256 + lo;
  return u >= 32768 ? u - 65536 : u;
}

// One plane: both IDCT passes block by block; res (Int32Array W*H) receives (t + 128) / 256
function idctPlane(coef, W, H, isChroma, qscale, intra, mbw, qm, res) {
  const X = new Int32Array(8), v = new Int32Array(8), t = new Int32Array(8);
  const scratch = new Int16Array(64);                      // one block of the pass-1 output, [z][n]
  for (let R = 0; R < H >> 3; R++) {
    for (let Q = 0; Q < W >> 3; Q++) {
      const mb = isChroma ? R * mbw + Q : (R >> 1) * mbw + (Q >> 1);
      const ag = intra[mb] > 0 ? 1 : 0, q = qscale[mb];
      const base = 8 * R * W + 8 * Q;
      for (let z = 0; z < 8; z++) {                        // pass 1: column z (mpeg1video.js:20-24)
        for (let i = 0; i < 8; i++) X[i] = coef[base + i * W + z];
        const dc = X[0];
        for (let i = 0; i < 8; i++) {
          if (X[i] === 0) continue;
          let x = 2 * X[i];
          const O = qm[(ag ? 0 : 64) + i * 8 + z];
          if (ag === 0) x += x < 0 ? -1 : 1;
          let f = Math.floor(x * q * O / 16);              // floor (D3); exact in a double
          if ((f & 1) === 0) f -= f > 0 ? 1 : -1;          // f == 0 becomes +1
          if (f > 2047) f = 2047;
          if (f < -2048) f = -2048;
          X[i] = f * PREMULTIPLIER[i * 8 + z];
        }
        if (z === 0 && ag === 1) X[0] = dc * 256;
        butterfly8(X, v);
        for (let n = 0; n < 8; n++)
          scratch[z * 8 + n] = handoffStore(Math.floor(Math.fround(Math.fround(v[n]) * Y04)));
      }
      for (let a = 0; a < 8; a++) {                        // pass 2: row a (mpeg1video.js:24-27)
        for (let i = 0; i < 8; i++) X[i] = Math.fround(Math.fround(scratch[i * 8 + a]) / Y04) | 0;
        butterfly8(X, t);
        const o = base + a * W;
        for (let m = 0; m < 8; m++) res[o + m] = ((t[m] + 128) / 256) | 0;
      }
    }
  }
}

// reference sample with the texel-granular CLAMP_TO_EDGE of the GL path (jsv.js:216-217, _p())
function refPx(ref, off, W, H, x, y) {
  let tx = x >> 2;
  if (tx < 0) tx = 0;
  if (tx > (W >> 2) - 1) tx = (W >> 2) - 1;
  if (y < 0) y = 0;
  if (y > H - 1) y = H - 1;
  return ref[off + y * W + 4 * tx + (x & 3)];
}

// prediction of one plane, macroblock by macroblock (copyMacroblock's cases: jsv.js:895-1129)
function predictPlane(ref, off, W, H, isChroma, mv, mbw, pred) {
  const s = isChroma ? 8 : 16;
  for (let my = 0; my < H / s; my++) {
    for (let mx = 0; mx < W / s; mx++) {
      const mb = my * mbw + mx;
      let h = mv[2 * mb], v = mv[2 * mb + 1];
      if (isChroma) { h = (h / 2) | 0; v = (v / 2) | 0; }  // toward zero, then floor / parity
      const ax = h >> 1, ay = v >> 1, oh = h & 1, ov = v & 1;
      for (let y = my * s; y < my * s + s; y++) {
        for (let x = mx * s; x < mx * s + s; x++) {
          const a = refPx(ref, off, W, H, x + ax, y + ay);
          let p;
          if (oh && ov) p = (a + refPx(ref, off, W, H, x + ax + 1, y + ay) + refPx(ref, off, W, H, x + ax, y + ay + 1) +
                             refPx(ref, off, W, H, x + ax + 1, y + ay + 1) + 2) >> 2;
          else if (oh) p = (a + refPx(ref, off, W, H, x + ax + 1, y + ay) + 1) >> 1;
          else if (ov) p = (a + refPx(ref, off, W, H, x + ax, y + ay + 1) + 1) >> 1;
          else p = a;
          pred[y * W + x] = p;
        }
      }
    }
  }
}

/*
 * One picture.  t: {type, coefY, coefCb, coefCr (Int16Array), qscale, intra, repadd, mbDir (Uint8Array),
 * mvFwd, mvBwd (Int16Array)}; qm: Uint8Array(128) intra then non-intra; refFwd / refBwd / out:
 * Uint8Array [Y | Cb | Cr].
 */
function decodePicture(t, cw, ch, qm, refFwd, refBwd, out) {
  const mbw = cw >> 4;
  const coef = [t.coefY, t.coefCb, t.coefCr];
  let off = 0;
  for (let comp = 0; comp < 3; comp++) {
    const W = comp ? cw >> 1 : cw, H = comp ? ch >> 1 : ch, isc = comp !== 0, s = isc ? 8 : 16;
    const res = new Int32Array(W * H);
    idctPlane(coef[comp], W, H, isc, t.qscale, t.intra, mbw, qm, res);
    let pf = null, pb = null;
    if (t.type !== 1) { pf = new Uint8Array(W * H); predictPlane(refFwd, off, W, H, isc, t.mvFwd, mbw, pf); }
    if (t.type === 3) { pb = new Uint8Array(W * H); predictPlane(refBwd, off, W, H, isc, t.mvBwd, mbw, pb); }
    for (let y = 0; y < H; y++) {
      for (let x = 0; x < W; x++) {
        const i = y * W + x, mb = ((y / s) | 0) * mbw + ((x / s) | 0);
        let p = 0;
        if (t.type === 2) p = t.repadd[mb] >= 128 ? 0 : pf[i];
        else if (t.type === 3) {
          const d = t.mbDir[mb] & 3;
          if (t.repadd[mb] >= 128 || d === 0) p = 0;
          else if (d === 1) p = pf[i];
          else if (d === 2) p = pb[i];
          else p = (pf[i] + pb[i] + 1) >> 1;
        }
        const v = res[i] + p;
        out[off + i] = v < 0 ? 0 : v > 255 ? 255 : v;
      }
    }
    off += W * H;
  }
}

// player/easybits.player.js:2674-2785, index progression restated literally (odd-width drift included)
function ycbcrToRgba(planes, codedW, codedH, frameW, frameH) {
  const rgba = new Uint8ClampedArray(frameW * frameH * 4).fill(255);
  const pY = planes.subarray(0, codedW * codedH), pCb = planes.subarray(codedW * codedH, codedW * codedH * 5 / 4);
  const pCr = planes.subarray(codedW * codedH * 5 / 4);
  const halfW = codedW >> 1;
  let yIndex1 = 0, yIndex2 = codedW, cIndex = 0, rgbaIndex1 = 0, rgbaIndex2 = frameW * 4;
  const yNext2Lines = codedW + (codedW - frameW), cNextLine = halfW - (frameW >> 1), rgbaNext2Lines = frameW * 4;
  const cols = frameW >> 1, rows = frameH >> 1;
  for (let row = 0; row < rows; row++) {
    for (let col = 0; col < cols; col++) {
      const cb = pCb[cIndex], cr = pCr[cIndex];
      cIndex++;
      const yuvr = cr - 128, yuvb = cb - 128;
      const r = yuvr * 1.59603, g = (-0.81297 * yuvr) - (0.39176 * yuvb), b = yuvb * 2.01723;
      for (let k = 0; k < 2; k++) {
        const ys = (pY[yIndex1++] - 16) * 1.16438;
        rgba[rgbaIndex1] = r + ys; rgba[rgbaIndex1 + 1] = g + ys; rgba[rgbaIndex1 + 2] = b + ys;
        rgbaIndex1 += 4;
      }
      for (let k = 0; k < 2; k++) {
        const ys = (pY[yIndex2++] - 16) * 1.16438;
        rgba[rgbaIndex2] = r + ys; rgba[rgbaIndex2 + 1] = g + ys; rgba[rgbaIndex2 + 2] = b + ys;
        rgbaIndex2 += 4;
      }
    }
    yIndex1 += yNext2Lines; yIndex2 += yNext2Lines;
    rgbaIndex1 += rgbaNext2Lines; rgbaIndex2 += rgbaNext2Lines;
    cIndex += cNextLine;
  }
  return rgba;
}

module.exports = { decodePicture, ycbcrToRgba, butterfly8, handoffStore };
