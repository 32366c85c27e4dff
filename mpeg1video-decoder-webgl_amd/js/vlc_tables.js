'use strict';
/*
 * vlc_tables.js -- ISO/IEC 11172-2 Annex B variable-length codes as (value, code, length)
 * lists, expanded once into flat multi-bit lookup tables (peek N bits -> symbol + length)
 * instead of the bit-serial tree walk of the reference (decoders/jsv.js:1593-1599,
 * tables :1815-2429).  Same symbols, same values the reference's readCode() returns.
 */

function buildLookup(entries, maxLen) {
  // entries: [value, code, len]; table[peek(maxLen)] = (len << 16) | (value & 0xffff); 0 = invalid
  const table = new Int32Array(1 << maxLen);
  for (const [value, code, len] of entries) {
    const shift = maxLen - len;
    const base = code << shift;
    const packed = (len << 16) | (value & 0xffff);
    for (let i = 0; i < (1 << shift); i++) table[base + i] = packed;
  }
  return { table, maxLen };
}

// Table B.1: macroblock_address_increment 1..33, 34 = stuffing, 35 = escape
const MBA_CODES = [[0x1, 1], [0x3, 3], [0x2, 3], [0x3, 4], [0x2, 4], [0x3, 5], [0x2, 5], [0x7, 7], [0x6, 7], [0xb, 8],
  [0xa, 8], [0x9, 8], [0x8, 8], [0x7, 8], [0x6, 8], [0x17, 10], [0x16, 10], [0x15, 10], [0x14, 10], [0x13, 10],
  [0x12, 10], [0x23, 11], [0x22, 11], [0x21, 11], [0x20, 11], [0x1f, 11], [0x1e, 11], [0x1d, 11], [0x1c, 11],
  [0x1b, 11], [0x1a, 11], [0x19, 11], [0x18, 11]];
const MBA = buildLookup(MBA_CODES.map(([c, l], i) => [i + 1, c, l]).concat([[34, 0xf, 11], [35, 0x8, 11]]), 11);

// Table B.2: macroblock_type; value = flags 0x10 quant | 0x08 fwd | 0x04 bwd | 0x02 pattern | 0x01 intra
const MBTYPE = [null,
  buildLookup([[0x01, 0x1, 1], [0x11, 0x1, 2]], 2),
  buildLookup([[0x0a, 0x1, 1], [0x02, 0x1, 2], [0x08, 0x1, 3], [0x01, 0x3, 5], [0x1a, 0x2, 5], [0x12, 0x1, 5], [0x11, 0x1, 6]], 6),
  buildLookup([[0x0c, 0x2, 2], [0x0e, 0x3, 2], [0x04, 0x2, 3], [0x06, 0x3, 3], [0x08, 0x2, 4], [0x0a, 0x3, 4], [0x01, 0x3, 5],
    [0x1e, 0x2, 5], [0x1a, 0x3, 6], [0x16, 0x2, 6], [0x11, 0x1, 6]], 6)];

// Table B.3: coded_block_pattern, index = cbp
const CBP_CODES = [[0x1, 9], [0xb, 5], [0x9, 5], [0xd, 6], [0xd, 4], [0x17, 7], [0x13, 7], [0x1f, 8], [0xc, 4], [0x16, 7],
  [0x12, 7], [0x1e, 8], [0x13, 5], [0x1b, 8], [0x17, 8], [0x13, 8], [0xb, 4], [0x15, 7], [0x11, 7], [0x1d, 8],
  [0x11, 5], [0x19, 8], [0x15, 8], [0x11, 8], [0xf, 6], [0xf, 8], [0xd, 8], [0x3, 9], [0xf, 5], [0xb, 8],
  [0x7, 8], [0x7, 9], [0xa, 4], [0x14, 7], [0x10, 7], [0x1c, 8], [0xe, 6], [0xe, 8], [0xc, 8], [0x2, 9],
  [0x10, 5], [0x18, 8], [0x14, 8], [0x10, 8], [0xe, 5], [0xa, 8], [0x6, 8], [0x6, 9], [0x12, 5], [0x1a, 8],
  [0x16, 8], [0x12, 8], [0xd, 5], [0x9, 8], [0x5, 8], [0x5, 9], [0xc, 5], [0x8, 8], [0x4, 8], [0x4, 9],
  [0x7, 3], [0xa, 5], [0x8, 5], [0xc, 6]];
const CBP = buildLookup(CBP_CODES.map(([c, l], i) => [i, c, l]).slice(1), 9);

// Table B.4: motion code magnitude 0..16; sign bit follows when non-zero (value = magnitude)
const MOTION_CODES = [[0x1, 1], [0x1, 2], [0x1, 3], [0x1, 4], [0x3, 6], [0x5, 7], [0x4, 7], [0x3, 7], [0xb, 9], [0xa, 9],
  [0x9, 9], [0x11, 10], [0x10, 10], [0xf, 10], [0xe, 10], [0xd, 10], [0xc, 10]];
const MOTION = buildLookup(MOTION_CODES.map(([c, l], i) => [i, c, l]), 10);

// Table B.5a/b: dct_dc_size
const DC_LUM = buildLookup([[0, 0x4, 3], [1, 0x0, 2], [2, 0x1, 2], [3, 0x5, 3], [4, 0x6, 3], [5, 0xe, 4], [6, 0x1e, 5],
  [7, 0x3e, 6], [8, 0x7e, 7]], 7);
const DC_CHR = buildLookup([[0, 0x0, 2], [1, 0x1, 2], [2, 0x2, 2], [3, 0x6, 3], [4, 0xe, 4], [5, 0x1e, 5], [6, 0x3e, 6],
  [7, 0x7e, 7], [8, 0xfe, 8]], 8);

// Table B.5c-g: dct coefficients, codes without the sign bit; value = (run << 8) | level,
// 0xffff = escape, 0x0001 = '1'/'11' (run 0 level 1; doubles as EOB '10', handled by the caller
// exactly like decoders/jsv.js:1405-1407)
const COEF_CODES = [
  [0x3, 2], [0x4, 4], [0x5, 5], [0x6, 7], [0x26, 8], [0x21, 8], [0xa, 10], [0x1d, 12], [0x18, 12], [0x13, 12],
  [0x10, 12], [0x1a, 13], [0x19, 13], [0x18, 13], [0x17, 13], [0x1f, 14], [0x1e, 14], [0x1d, 14], [0x1c, 14],
  [0x1b, 14], [0x1a, 14], [0x19, 14], [0x18, 14], [0x17, 14], [0x16, 14], [0x15, 14], [0x14, 14], [0x13, 14],
  [0x12, 14], [0x11, 14], [0x10, 14], [0x18, 15], [0x17, 15], [0x16, 15], [0x15, 15], [0x14, 15], [0x13, 15],
  [0x12, 15], [0x11, 15], [0x10, 15],
  [0x3, 3], [0x6, 6], [0x25, 8], [0xc, 10], [0x1b, 12], [0x16, 13], [0x15, 13], [0x1f, 15], [0x1e, 15], [0x1d, 15],
  [0x1c, 15], [0x1b, 15], [0x1a, 15], [0x19, 15], [0x13, 16], [0x12, 16], [0x11, 16], [0x10, 16],
  [0x5, 4], [0x4, 7], [0xb, 10], [0x14, 12], [0x14, 13],
  [0x7, 5], [0x24, 8], [0x1c, 12], [0x13, 13],
  [0x6, 5], [0xf, 10], [0x12, 12],
  [0x7, 6], [0x9, 10], [0x12, 13],
  [0x5, 6], [0x1e, 12], [0x14, 16],
  [0x4, 6], [0x15, 12], [0x7, 7], [0x11, 12], [0x5, 7], [0x11, 13], [0x27, 8], [0x10, 13],
  [0x23, 8], [0x1a, 16], [0x22, 8], [0x19, 16], [0x20, 8], [0x18, 16], [0xe, 10], [0x17, 16], [0xd, 10], [0x16, 16],
  [0x8, 10], [0x15, 16],
  [0x1f, 12], [0x1a, 12], [0x19, 12], [0x17, 12], [0x16, 12], [0x1f, 13], [0x1e, 13], [0x1d, 13], [0x1c, 13],
  [0x1b, 13], [0x1f, 16], [0x1e, 16], [0x1d, 16], [0x1c, 16], [0x1b, 16]];
const RUN_LEVELS = [];
for (let l = 1; l <= 40; l++) RUN_LEVELS.push([0, l]);
for (let l = 1; l <= 18; l++) RUN_LEVELS.push([1, l]);
for (let l = 1; l <= 5; l++) RUN_LEVELS.push([2, l]);
for (let l = 1; l <= 4; l++) RUN_LEVELS.push([3, l]);
for (let r = 4; r <= 6; r++) for (let l = 1; l <= 3; l++) RUN_LEVELS.push([r, l]);
for (let r = 7; r <= 16; r++) for (let l = 1; l <= 2; l++) RUN_LEVELS.push([r, l]);
for (let r = 17; r <= 31; r++) RUN_LEVELS.push([r, 1]);
if (RUN_LEVELS.length !== COEF_CODES.length) throw new Error('coefficient table size');
const coefEntries = COEF_CODES.map(([c, l], i) => [(RUN_LEVELS[i][0] << 8) | RUN_LEVELS[i][1], c, l]);
coefEntries[0] = [0x0001, 0x1, 1];            // '1': first-coefficient form; '1' + next bit decides EOB / '11'
coefEntries.push([0xffff, 0x1, 6]);           // escape
const COEF = buildLookup(coefEntries, 16);

const ZIG_ZAG = new Uint8Array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
  27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45,
  38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]);

const PICTURE_RATE = [0, 23.976, 24, 25, 29.97, 30, 50, 59.94, 60, 15, 5, 10, 12, 15, 0, 0];   // decoders/jsv.js:1762-1765

const DEFAULT_INTRA_QUANT_MATRIX = new Uint8Array([
  8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
  22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83]);
const DEFAULT_NON_INTRA_QUANT_MATRIX = new Uint8Array(64).fill(16);

module.exports = { MBA, MBTYPE, CBP, MOTION, DC_LUM, DC_CHR, COEF, ZIG_ZAG, PICTURE_RATE,
  DEFAULT_INTRA_QUANT_MATRIX, DEFAULT_NON_INTRA_QUANT_MATRIX };
