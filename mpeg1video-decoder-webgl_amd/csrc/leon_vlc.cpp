// leon_vlc.cpp -- native bitstream front end behind include/leon_vlc.h (no GPU code).
//
// Same stream semantics as js/jsv_decoder.js (the JavaScript mirror of the reference's parser,
// decoders/jsv.js), restated for throughput:
//   - every variable-length code is one table lookup on the next 6..16 bits
//     (the reference walks a binary tree one bit at a time, jsv.js:1593-1599)
//   - the slices of a picture are located by their start codes first and then decoded
//     concurrently on a small thread pool: a slice carries its own predictors, quantiser scale and
//     macroblock address (jsv.js:683-706), and writes only its own macroblocks
//   - coefficients leave as sparse (group, tile offset, level) entries, bucketed per 64x8 group by
//     a counting sort, instead of dense planes
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../../include/leon_vlc.h"

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- ISO/IEC 11172-2 Annex B tables as (value, code, length), expanded to flat lookups ----------

struct Lookup {
    std::vector<int32_t> t;      // (len << 16) | value; 0 = invalid
    int max_len = 0;
    void init(int ml) { max_len = ml; t.assign((size_t)1 << ml, 0); }
    void add(int value, unsigned code, int len)
    {
        const int shift = max_len - len;
        const int32_t packed = (len << 16) | (value & 0xffff);
        for (unsigned i = 0; i < (1u << shift); i++) t[((size_t)code << shift) + i] = packed;
    }
};

struct CL { unsigned code; int len; };

const CL kMba[33] = {{0x1, 1}, {0x3, 3}, {0x2, 3}, {0x3, 4}, {0x2, 4}, {0x3, 5}, {0x2, 5}, {0x7, 7}, {0x6, 7}, {0xb, 8},
    {0xa, 8}, {0x9, 8}, {0x8, 8}, {0x7, 8}, {0x6, 8}, {0x17, 10}, {0x16, 10}, {0x15, 10}, {0x14, 10}, {0x13, 10},
    {0x12, 10}, {0x23, 11}, {0x22, 11}, {0x21, 11}, {0x20, 11}, {0x1f, 11}, {0x1e, 11}, {0x1d, 11}, {0x1c, 11},
    {0x1b, 11}, {0x1a, 11}, {0x19, 11}, {0x18, 11}};
const CL kCbp[64] = {{0x1, 9}, {0xb, 5}, {0x9, 5}, {0xd, 6}, {0xd, 4}, {0x17, 7}, {0x13, 7}, {0x1f, 8}, {0xc, 4}, {0x16, 7},
    {0x12, 7}, {0x1e, 8}, {0x13, 5}, {0x1b, 8}, {0x17, 8}, {0x13, 8}, {0xb, 4}, {0x15, 7}, {0x11, 7}, {0x1d, 8},
    {0x11, 5}, {0x19, 8}, {0x15, 8}, {0x11, 8}, {0xf, 6}, {0xf, 8}, {0xd, 8}, {0x3, 9}, {0xf, 5}, {0xb, 8},
    {0x7, 8}, {0x7, 9}, {0xa, 4}, {0x14, 7}, {0x10, 7}, {0x1c, 8}, {0xe, 6}, {0xe, 8}, {0xc, 8}, {0x2, 9},
    {0x10, 5}, {0x18, 8}, {0x14, 8}, {0x10, 8}, {0xe, 5}, {0xa, 8}, {0x6, 8}, {0x6, 9}, {0x12, 5}, {0x1a, 8},
    {0x16, 8}, {0x12, 8}, {0xd, 5}, {0x9, 8}, {0x5, 8}, {0x5, 9}, {0xc, 5}, {0x8, 8}, {0x4, 8}, {0x4, 9},
    {0x7, 3}, {0xa, 5}, {0x8, 5}, {0xc, 6}};
const CL kMotion[17] = {{0x1, 1}, {0x1, 2}, {0x1, 3}, {0x1, 4}, {0x3, 6}, {0x5, 7}, {0x4, 7}, {0x3, 7}, {0xb, 9}, {0xa, 9},
    {0x9, 9}, {0x11, 10}, {0x10, 10}, {0xf, 10}, {0xe, 10}, {0xd, 10}, {0xc, 10}};
const CL kCoef[111] = {
    {0x3, 2}, {0x4, 4}, {0x5, 5}, {0x6, 7}, {0x26, 8}, {0x21, 8}, {0xa, 10}, {0x1d, 12}, {0x18, 12}, {0x13, 12},
    {0x10, 12}, {0x1a, 13}, {0x19, 13}, {0x18, 13}, {0x17, 13}, {0x1f, 14}, {0x1e, 14}, {0x1d, 14}, {0x1c, 14},
    {0x1b, 14}, {0x1a, 14}, {0x19, 14}, {0x18, 14}, {0x17, 14}, {0x16, 14}, {0x15, 14}, {0x14, 14}, {0x13, 14},
    {0x12, 14}, {0x11, 14}, {0x10, 14}, {0x18, 15}, {0x17, 15}, {0x16, 15}, {0x15, 15}, {0x14, 15}, {0x13, 15},
    {0x12, 15}, {0x11, 15}, {0x10, 15},
    {0x3, 3}, {0x6, 6}, {0x25, 8}, {0xc, 10}, {0x1b, 12}, {0x16, 13}, {0x15, 13}, {0x1f, 15}, {0x1e, 15}, {0x1d, 15},
    {0x1c, 15}, {0x1b, 15}, {0x1a, 15}, {0x19, 15}, {0x13, 16}, {0x12, 16}, {0x11, 16}, {0x10, 16},
    {0x5, 4}, {0x4, 7}, {0xb, 10}, {0x14, 12}, {0x14, 13},
    {0x7, 5}, {0x24, 8}, {0x1c, 12}, {0x13, 13},
    {0x6, 5}, {0xf, 10}, {0x12, 12},
    {0x7, 6}, {0x9, 10}, {0x12, 13},
    {0x5, 6}, {0x1e, 12}, {0x14, 16},
    {0x4, 6}, {0x15, 12}, {0x7, 7}, {0x11, 12}, {0x5, 7}, {0x11, 13}, {0x27, 8}, {0x10, 13},
    {0x23, 8}, {0x1a, 16}, {0x22, 8}, {0x19, 16}, {0x20, 8}, {0x18, 16}, {0xe, 10}, {0x17, 16}, {0xd, 10}, {0x16, 16},
    {0x8, 10}, {0x15, 16},
    {0x1f, 12}, {0x1a, 12}, {0x19, 12}, {0x17, 12}, {0x16, 12}, {0x1f, 13}, {0x1e, 13}, {0x1d, 13}, {0x1c, 13},
    {0x1b, 13}, {0x1f, 16}, {0x1e, 16}, {0x1d, 16}, {0x1c, 16}, {0x1b, 16}};

const uint8_t kZigZag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
    27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45,
    38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const double kPictureRate[16] = {0, 23.976, 24, 25, 29.97, 30, 50, 59.94, 60, 15, 5, 10, 12, 15, 0, 0};   // jsv.js:1762-1765
const uint8_t kDefaultIntra[64] = {
    8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};

struct Tables {
    Lookup mba, mbtype[4], cbp, motion, dc_lum, dc_chr, coef;
    int32_t coef8[256];          // the coefficient codes of at most 8 bits (most of them): 1 KB, stays in L1
    // Every symbol AFTER the first of a block whose code, sign bit included, fits 12 bits, resolved by ONE
    // lookup on the next 12 bits: bits 0..6 length (sign included), bit 7 = end of block ('10'),
    // bits 8..15 run, bits 16..31 the signed level.  0 = longer code or escape: the general path.  16 KB.
    uint32_t fast12[4096];
    // The same for the FIRST symbol of a block, where '1s' is run 0, level +-1 and no end of block exists.
    uint32_t first12[4096];
    // motion_code with its sign bit in one lookup on 11 bits: (len << 16) | (code + 16); 0 = invalid
    int32_t motion_s[2048];
    Tables()
    {
        mba.init(11);
        for (int i = 0; i < 33; i++) mba.add(i + 1, kMba[i].code, kMba[i].len);
        mba.add(34, 0xf, 11);                                  // stuffing
        mba.add(35, 0x8, 11);                                  // escape
        // macroblock_type flags: 0x10 quant | 0x08 fwd | 0x04 bwd | 0x02 pattern | 0x01 intra
        mbtype[1].init(2);
        mbtype[1].add(0x01, 0x1, 1); mbtype[1].add(0x11, 0x1, 2);
        mbtype[2].init(6);
        const int p[7][3] = {{0x0a, 0x1, 1}, {0x02, 0x1, 2}, {0x08, 0x1, 3}, {0x01, 0x3, 5}, {0x1a, 0x2, 5}, {0x12, 0x1, 5}, {0x11, 0x1, 6}};
        for (auto& e : p) mbtype[2].add(e[0], (unsigned)e[1], e[2]);
        mbtype[3].init(6);
        const int b[11][3] = {{0x0c, 0x2, 2}, {0x0e, 0x3, 2}, {0x04, 0x2, 3}, {0x06, 0x3, 3}, {0x08, 0x2, 4}, {0x0a, 0x3, 4},
                              {0x01, 0x3, 5}, {0x1e, 0x2, 5}, {0x1a, 0x3, 6}, {0x16, 0x2, 6}, {0x11, 0x1, 6}};
        for (auto& e : b) mbtype[3].add(e[0], (unsigned)e[1], e[2]);
        cbp.init(9);
        for (int i = 1; i < 64; i++) cbp.add(i, kCbp[i].code, kCbp[i].len);
        motion.init(10);
        for (int i = 0; i < 17; i++) motion.add(i, kMotion[i].code, kMotion[i].len);
        for (unsigned p = 0; p < 2048; p++) {
            const int32_t e = motion.t[p >> 1];                    // the 10-bit table of the unsigned codes
            int32_t m = 0;
            if (e != 0) {
                const int len = e >> 16, code = e & 0xffff;
                if (code == 0) m = (len << 16) | 16;               // '1': no sign bit follows
                else m = ((len + 1) << 16) | (((p >> (10 - len)) & 1) ? 16 - code : 16 + code);
            }
            motion_s[p] = m;
        }
        dc_lum.init(7);
        const int dl[9][2] = {{0x4, 3}, {0x0, 2}, {0x1, 2}, {0x5, 3}, {0x6, 3}, {0xe, 4}, {0x1e, 5}, {0x3e, 6}, {0x7e, 7}};
        for (int i = 0; i < 9; i++) dc_lum.add(i, (unsigned)dl[i][0], dl[i][1]);
        dc_chr.init(8);
        const int dc[9][2] = {{0x0, 2}, {0x1, 2}, {0x2, 2}, {0x6, 3}, {0xe, 4}, {0x1e, 5}, {0x3e, 6}, {0x7e, 7}, {0xfe, 8}};
        for (int i = 0; i < 9; i++) dc_chr.add(i, (unsigned)dc[i][0], dc[i][1]);
        // dct coefficients without the sign bit; value = (run << 8) | level, 0xffff = escape,
        // 0x0001 = '1' (run 0 level 1 in first position / with the next bit: '10' EOB, '11' 0/1)
        coef.init(16);
        int runs[111], levels[111], k = 0;
        for (int l = 1; l <= 40; l++) { runs[k] = 0; levels[k++] = l; }
        for (int l = 1; l <= 18; l++) { runs[k] = 1; levels[k++] = l; }
        for (int l = 1; l <= 5; l++) { runs[k] = 2; levels[k++] = l; }
        for (int l = 1; l <= 4; l++) { runs[k] = 3; levels[k++] = l; }
        for (int r = 4; r <= 6; r++) for (int l = 1; l <= 3; l++) { runs[k] = r; levels[k++] = l; }
        for (int r = 7; r <= 16; r++) for (int l = 1; l <= 2; l++) { runs[k] = r; levels[k++] = l; }
        for (int r = 17; r <= 31; r++) { runs[k] = r; levels[k++] = 1; }
        for (int i = 1; i < 111; i++) coef.add((runs[i] << 8) | levels[i], kCoef[i].code, kCoef[i].len);
        coef.add(0x0001, 0x1, 1);
        coef.add(0xffff, 0x1, 6);
        for (int i = 0; i < 256; i++) {
            const int32_t e = coef.t[(size_t)i << 8];
            coef8[i] = (e != 0 && (e >> 16) <= 8) ? e : 0;
        }
        for (unsigned p = 0; p < 4096; p++) {
            uint32_t f = 0;
            if ((p >> 10) == 2) f = 0x80u | 2u;                                   // '10': end of block
            else if ((p >> 10) == 3) f = (uint32_t)(((p >> 9) & 1) ? -1 : 1) << 16 | 3u;   // '11s': run 0, level +-1
            else {
                const int32_t e = coef.t[(size_t)p << 4];
                const int len = e >> 16, cf = e & 0xffff;
                if (e != 0 && cf != 0xffff && cf != 0x0001 && len + 1 <= 12) {
                    int level = cf & 0xff;
                    if ((p >> (11 - len)) & 1) level = -level;
                    f = ((uint32_t)(uint16_t)(int16_t)level << 16) | ((uint32_t)(cf >> 8) << 8) | (uint32_t)(len + 1);
                }
            }
            fast12[p] = f;
            first12[p] = (p >> 11) ? ((uint32_t)(uint16_t)(int16_t)(((p >> 10) & 1) ? -1 : 1) << 16) | 2u : f;
        }
    }
};
const Tables& tables()
{
    static const Tables t;
    return t;
}

// ---- bit reader over a zero-padded copy of the stream ------------------------------------------

struct Bits {
    const uint8_t* b = nullptr;
    size_t nbytes = 0;           // real length (the buffer has 16 readable bytes more)
    size_t pos = 0;              // in bits
    bool bad = false;

    uint32_t peek(int n) const   // n <= 24
    {
        const uint8_t* p = b + (pos >> 3);
        const uint32_t v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        return (v << (pos & 7)) >> (32 - n);
    }
    uint32_t get(int n)          // n <= 32
    {
        if (n == 0) return 0;
        if (n > 24) { const uint32_t hi = get(n - 16); return (hi << 16) | get(16); }
        if (pos + (size_t)n > nbytes * 8 + 32) { bad = true; return 0; }
        const uint32_t v = peek(n);
        pos += (size_t)n;
        return v;
    }
    void skip(size_t n) { pos += n; }
    int vlc(const Lookup& t)
    {
        if ((pos >> 3) >= nbytes) { bad = true; return 0; }
        const int32_t e = t.t[peek(t.max_len)];
        if (e == 0) { bad = true; return 0; }
        pos += (size_t)(e >> 16);
        return e & 0xffff;
    }
    // byte-aligned scan for 00 00 01 xx; returns xx with pos behind it, or -1 at the end.
    // The next 00 00 01 xx at or behind the current byte, 16 bytes at a time: (b[j] == 0) & (b[j + 1] == 0) & (b[j + 2] == 1)
    // as three unaligned loads and compares.  (Byte by byte the picture layer scan of the GPU parser's host side spent
    // most of its 80 us per 1080p picture here; memchr for the 01s -- one byte in 60 of the 1080p test streams -- brought
    // it to 50 us, this to a third of that.)
    int next_start_code()
    {
        size_t j = (pos + 7) >> 3;                          // candidate position of the first zero
#if defined(__SSE2__)
        const __m128i zero = _mm_setzero_si128(), one = _mm_set1_epi8(1);
        while (j + 18 < nbytes) {                           // reads b[j .. j + 17]; xx of a hit at j + 3 <= j + 18 < nbytes
            const __m128i a = _mm_loadu_si128((const __m128i*)(b + j)), c = _mm_loadu_si128((const __m128i*)(b + j + 1)),
                          d = _mm_loadu_si128((const __m128i*)(b + j + 2));
            const int m = _mm_movemask_epi8(_mm_and_si128(_mm_and_si128(_mm_cmpeq_epi8(a, zero), _mm_cmpeq_epi8(c, zero)), _mm_cmpeq_epi8(d, one)));
            if (m) {
                j += (size_t)__builtin_ctz((unsigned)m);
                pos = (j + 4) << 3;
                return b[j + 3];
            }
            j += 16;
        }
#endif
        for (; j + 3 < nbytes; j++)
            if (b[j] == 0 && b[j + 1] == 0 && b[j + 2] == 1) { pos = (j + 4) << 3; return b[j + 3]; }
        pos = nbytes * 8;
        return -1;
    }
    bool next_bits_are_start_code() const      // decoders/jsv.js:1710-1760
    {
        const size_t i = (pos + 7) >> 3;
        if (i + 2 >= nbytes) return true;
        return b[i] == 0 && b[i + 1] == 0 && b[i + 2] == 1;
    }
};

// The slice decoder's view of the stream: a 64-bit window `w` that holds the bits from `pos` on, left aligned,
// `avail` of them valid; reloaded (8 bytes, byte swapped) only when a caller needs more than are left.  Every
// syntax element below the slice header -- macroblock header, vectors, DC sizes, coefficients -- is read from it
// with shifts and one table lookup; nothing recomputes a byte address per element.
struct Win {
    const uint8_t* b = nullptr;
    size_t nbytes = 0;           // real length (the buffer has 16 readable bytes more, all zero)
    size_t pos = 0;              // in bits
    uint64_t w = 0;
    int avail = 0;
    bool bad = false;

    void refill()
    {
        if ((pos >> 3) >= nbytes) { bad = true; w = 0; avail = 64; return; }       // zeros: no table loops on them
        memcpy(&w, b + (pos >> 3), 8);
        w = __builtin_bswap64(w) << (pos & 7);
        avail = 64 - (int)(pos & 7);
    }
    void need(int n) { if (avail < n) refill(); }            // n <= 57
    uint32_t peek(int n) const { return (uint32_t)(w >> (64 - n)); }      // 1 <= n <= 32, after need(n)
    void drop(int n) { w <<= n; avail -= n; pos += (size_t)n; }
    uint32_t get(int n)          // n <= 32
    {
        if (n == 0) return 0;
        need(n);
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    int vlc(const Lookup& t)     // max_len <= 16
    {
        need(t.max_len);
        const int32_t e = t.t[peek(t.max_len)];
        if (e == 0) { bad = true; return 0; }
        drop(e >> 16);
        return e & 0xffff;
    }
    bool next_bits_are_start_code() const      // decoders/jsv.js:1710-1760
    {
        const size_t i = (pos + 7) >> 3;
        if (i + 2 >= nbytes) return true;
        return b[i] == 0 && b[i + 1] == 0 && b[i + 2] == 1;
    }
};

enum { START_PICTURE = 0x00, START_SLICE_FIRST = 0x01, START_SLICE_LAST = 0xAF, START_USER_DATA = 0xB2,
       START_SEQUENCE_ES = 0xB3, START_SEQUENCE = 0xC3, START_EXTENSION = 0xB5, START_GOP = 0xB8, START_MAP = 0xC4 };

struct SliceJob { int code; size_t bitpos; };

// What one slice contributed to one macroblock row: four entry streams (upper and lower luma block
// row, Cb, Cr), each already ordered by group because a slice walks its macroblocks left to right,
// plus the entry count per group.  No sort is needed afterwards -- the streams are concatenated.
struct RowRun {                             // streams: 0/1 luma block rows, 2 Cb, 3 Cr, 4/5 the A plane's block rows (yuva)
    int mb_row = 0;
    std::vector<uint32_t> ent[6];              // sized ahead of `used` (a block appends up to 64 entries through a raw pointer)
    size_t used[6] = {0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> cnt[6];
};
struct alignas(128) SliceOut {             // one per slice; written by exactly one worker
    std::vector<RowRun> runs;
    size_t used = 0;
};

}  // namespace

struct leon_vlc_stream {
    std::vector<uint8_t> data;
    Bits r;
    leon_vlc_info info{};
    bool have_meta = false, sequence_started = false, skip_till_gop = true, ended = false, new_sequence = false;
    bool raw_es = false;
    std::vector<uint32_t> keymap;              // (byte offset, time code) pairs
    double ts_pending = 0;
    bool open_gop_pending = false;          // a GOP header with closed_gop = 0 was read; reported with the next picture
    int temporal_reference = 0;

    // per-picture state (decoders/jsv.js:583-676)
    int type = 0;
    int full_pel_fwd = 0, fwd_rsize = 0, fwd_f = 1, full_pel_bwd = 0, bwd_rsize = 0, bwd_f = 1;
    int mbw = 0, mbh = 0, mbsize = 0, gy = 0, gc = 0, n_y = 0, n_c = 0;
    int n_streams = 4;                          // 6 for a yuva stream (container flag `a`)
    std::vector<uint8_t> qscale, intra, repadd, mb_dir;
    std::vector<int16_t> mv_fwd, mv_bwd;
    std::vector<uint32_t> grp_off, entries, cursor;

    // thread pool
    int n_threads = 1;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::atomic<uint64_t> generation{0};
    std::atomic<bool> quit{false};
    std::atomic<int> busy{0};
    std::vector<SliceJob> jobs;
    std::atomic<size_t> next_job{0};
    std::vector<SliceOut> slice_out;           // indexed like jobs; buffers are reused from picture to picture

    // parse-ahead: while the caller works on the picture it was handed, a coordinator thread
    // already parses the next one into the other of two result sets
    struct Result {
        std::vector<uint32_t> grp_off, entries;
        std::vector<uint8_t> qscale, intra, repadd, mb_dir;
        std::vector<int16_t> mv_fwd, mv_bwd;
        leon_vlc_picture pic{};
        leon_vlc_info info{};
        int rc = 0;
        char err[256] = "";
    };
    Result results[2];
    leon_vlc_info info_out{};                  // what leon_vlc_get_info reports: the state at the last returned picture
    std::thread ahead;
    std::mutex amu;
    std::condition_variable acv;
    int a_request = -1;                        // result set to fill, -1 = none
    int a_inflight = -1;                       // result set being filled or filled and not yet handed out
    bool a_done = false, a_quit = false, a_eos = false;
    std::atomic<int> slice_error{0};
    char slice_err_text[160] = "";
    // leon_vlc_scan_picture: headers and slice positions only
    bool scan_only = false;
    bool scan_stream = false;    // leon_vlc_open_scan: the bytes are the caller's, leon_vlc_next_picture is refused
    std::vector<int32_t> scan_code;
    std::vector<uint64_t> scan_pos;
    uint64_t scan_end = 0;
    std::mutex err_mu;
};

namespace {

struct SliceCtx {
    const Tables* T = &tables();      // looked up once per slice: the function-local static's guard showed up in the profile
    leon_vlc_stream* s;
    Win r;
    SliceOut* sout;
    RowRun* run = nullptr;
    int mb_addr = 0, mb_row = 0, mb_col = 0, rc_addr = 0;      // (mb_row, mb_col) = position of address rc_addr
    bool slice_begin = true;
    int fw_h = 0, fw_v = 0, fw_h_prev = 0, fw_v_prev = 0, bw_h = 0, bw_v = 0, bw_h_prev = 0, bw_v_prev = 0, prev_dir = 0;
    int dc_y = 128, dc_cr = 128, dc_cb = 128, dc_a = 128, qs = 0;
    int mb_intra = 0, mot_fw = 0, mot_bw = 0;
    const char* err = nullptr;
};

int motion_component(SliceCtx& c, int prev, int rsize, int f)
{
    Win& r = c.r;
    r.need(11 + 8);                                            // code, sign and the residual (r_size <= 6) in one window
    const int32_t e = c.T->motion_s[r.peek(11)];
    if (e == 0) { r.bad = true; return prev; }
    r.drop(e >> 16);
    int code = (e & 0xffff) - 16, d;
    if (code != 0 && f != 1) {
        const int res = (int)r.peek(rsize);                    // rsize >= 1 here
        r.drop(rsize);
        d = (((code < 0 ? -code : code) - 1) << rsize) + res + 1;
        if (code < 0) d = -d;
    } else d = code;
    prev += d;
    if (prev > (f << 4) - 1) prev -= f << 5;
    else if (prev < -(f << 4)) prev += f << 5;
    return prev;
}

// decoders/jsv.js:831-893 (+ backward vectors)
void decode_motion_vectors(SliceCtx& c)
{
    leon_vlc_stream* s = c.s;
    if (c.mot_fw) {
        c.fw_h_prev = motion_component(c, c.fw_h_prev, s->fwd_rsize, s->fwd_f);
        c.fw_h = s->full_pel_fwd ? c.fw_h_prev * 2 : c.fw_h_prev;
        c.fw_v_prev = motion_component(c, c.fw_v_prev, s->fwd_rsize, s->fwd_f);
        c.fw_v = s->full_pel_fwd ? c.fw_v_prev * 2 : c.fw_v_prev;
    } else if (s->type == 2) {
        c.fw_h = c.fw_h_prev = 0;
        c.fw_v = c.fw_v_prev = 0;
    }
    if (c.mot_bw) {
        c.bw_h_prev = motion_component(c, c.bw_h_prev, s->bwd_rsize, s->bwd_f);
        c.bw_h = s->full_pel_bwd ? c.bw_h_prev * 2 : c.bw_h_prev;
        c.bw_v_prev = motion_component(c, c.bw_v_prev, s->bwd_rsize, s->bwd_f);
        c.bw_v = s->full_pel_bwd ? c.bw_v_prev * 2 : c.bw_v_prev;
    }
}

// decoders/jsv.js:1338-1525 (decodeBlockGL): raw levels, emitted as sparse entries
bool decode_block(SliceCtx& c, int block)
{
    Win& r = c.r;
    const Tables& T = *c.T;
    // stream of the current row run and group inside the row
    int stream;
    uint32_t grow, bq;
    if (block < 4 || block >= 6) {                           // luma, or the A component (blocks 6..9, placed like luma)
        const int lb = block < 4 ? block : block - 6;
        const int qb = c.mb_col * 2 + (lb & 1);
        stream = (lb >> 1) + (block < 4 ? 0 : 4);
        grow = (uint32_t)(qb >> 3);
        bq = (uint32_t)(qb & 7);
    } else {
        stream = block - 2;                                  // 2 = Cb, 3 = Cr
        grow = (uint32_t)(c.mb_col >> 3);
        bq = (uint32_t)(c.mb_col & 7);
    }
    const uint32_t boff = (bq * 16u) << 16;
    // entries go straight to the end of the run's stream: room for a whole block is made first
    RowRun& run = *c.run;
    std::vector<uint32_t>& ev = run.ent[stream];
    const size_t used = run.used[stream];
    if (ev.size() < used + 64) ev.resize(std::max(ev.size() * 2, used + 64 + 960));
    uint32_t* const out = ev.data() + used;
    int k = 0, n = 0;
    if (c.mb_intra) {
        int predictor, size;
        if (block < 4) { predictor = c.dc_y; size = r.vlc(T.dc_lum); }
        else if (block >= 6) { predictor = c.dc_a; size = r.vlc(T.dc_lum); }
        else { predictor = block == 4 ? c.dc_cr : c.dc_cb; size = r.vlc(T.dc_chr); }
        int dc = predictor;
        if (size > 0) {
            const int differential = (int)r.get(size);
            dc = (differential & (1 << (size - 1))) ? predictor + differential
                                                    : predictor + ((int)(0xffffffffu << size) | (differential + 1));
        }
        if (block < 4) c.dc_y = dc; else if (block >= 6) c.dc_a = dc; else if (block == 4) c.dc_cr = dc; else c.dc_cb = dc;
        if ((int16_t)dc != 0) out[k++] = boff | (uint16_t)(int16_t)dc;
        n = 1;
    }
    // The coefficient loop: one table lookup and one window update per coefficient (a symbol takes at most
    // 28 bits).  Same decisions as the reference's loop, jsv.js:1396-1443.  The window lives in locals here (the
    // entry stores could alias the struct's fields as far as the compiler knows) and goes back at every exit.
    const uint32_t* fast = n > 0 ? T.fast12 : T.first12;       // the first symbol of a block has no end-of-block code
    uint64_t w = r.w;
    int avail = r.avail;
    size_t pos = r.pos;
    const uint8_t* const bytes = r.b;
    const size_t end_byte = r.nbytes;
#define LEON_WIN_BACK() do { r.w = w; r.avail = avail; r.pos = pos; } while (0)
    for (;;) {
        if (avail < 28) {
            if ((pos >> 3) >= end_byte) { LEON_WIN_BACK(); r.bad = true; c.err = "bitstream ends inside a block"; return false; }
            memcpy(&w, bytes + (pos >> 3), 8);
            w = __builtin_bswap64(w) << (pos & 7);
            avail = 64 - (int)(pos & 7);
        }
        const uint32_t f = fast[w >> 52];
        const int flen = (int)(f & 0x7fu);
        if (flen) {
            w <<= flen;
            avail -= flen;
            pos += (size_t)flen;
            if (f & 0x80u) break;                             // end of block
            n += (int)((f >> 8) & 0xffu);
            if (n > 63) { LEON_WIN_BACK(); c.err = "coefficient index overflow"; return false; }
            const uint32_t z = kZigZag[n++];
            out[k++] = boff | (((z >> 3) * 128u + (z & 7u) * 2u) << 16) | (f >> 16);
            // (resolving the FOLLOWING symbol from the same 12 bits as well -- a second table -- was measured on
            // the GPU box's EPYC 9575F: 890-900 pictures/s per thread against 900-910 without; not kept)
            fast = T.fast12;
            continue;
        }
        // longer codes and escapes ('1...' never gets here: both tables resolve it)
        int32_t e = T.coef8[w >> 56];
        if (e == 0) {
            e = T.coef.t[w >> 48];
            if (e == 0) { LEON_WIN_BACK(); r.bad = true; c.err = "invalid coefficient code"; return false; }
        }
        const int len = e >> 16, coeff = e & 0xffff;
        int run_len, level, used_bits;
        if (coeff == 0xffff) {                               // escape: 6-bit run, 8- or 16-bit level
            run_len = (int)((w >> 52) & 63);
            level = (int)((w >> 44) & 255);
            used_bits = 20;
            if (level == 0) { level = (int)((w >> 36) & 255); used_bits = 28; }
            else if (level == 128) { level = (int)((w >> 36) & 255) - 256; used_bits = 28; }
            else if (level > 128) level -= 256;
        } else {
            run_len = coeff >> 8;
            level = coeff & 0xff;
            if ((w >> (63 - len)) & 1) level = -level;
            used_bits = len + 1;
        }
        w <<= used_bits;
        avail -= used_bits;
        pos += (size_t)used_bits;
        fast = T.fast12;
        n += run_len;
        if (n > 63) { LEON_WIN_BACK(); c.err = "coefficient index overflow"; return false; }
        const uint32_t z = kZigZag[n++];
        if (level != 0) out[k++] = boff | (((z >> 3) * 128u + (z & 7u) * 2u) << 16) | (uint16_t)(int16_t)level;
    }
    LEON_WIN_BACK();
#undef LEON_WIN_BACK
    run.used[stream] = used + (size_t)k;
    run.cnt[stream][grow] += (uint32_t)k;
    return !r.bad;
}

// decoders/jsv.js:725-828 (+ B pictures).  1 = macroblock read, 0 = error (c.err), 2 = the
// reference's silent return for an address past the picture (the slice loop goes on)
int decode_macroblock(SliceCtx& c)
{
    leon_vlc_stream* s = c.s;
    Win& r = c.r;
    const Tables& T = *c.T;
    const int type = s->type;
    int increment = 0, t = r.vlc(T.mba);
    while (t == 34 && !r.bad) t = r.vlc(T.mba);
    while (t == 35 && !r.bad) { increment += 33; t = r.vlc(T.mba); }
    if (r.bad) { c.err = "invalid macroblock address increment"; return 0; }
    increment += t;
    if (c.slice_begin) {
        c.slice_begin = false;
        c.mb_addr += increment;
    } else {
        if (c.mb_addr + increment >= s->mbsize) return 2;
        if (increment > 1) {
            c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
            if (type == 2) { c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0; }
        }
        while (increment > 1) {                                    // skipped macroblocks
            const int a = ++c.mb_addr;
            s->mv_fwd[2 * a] = (int16_t)c.fw_h;
            s->mv_fwd[2 * a + 1] = (int16_t)c.fw_v;
            if (type == 3) {
                s->mv_bwd[2 * a] = (int16_t)c.bw_h;
                s->mv_bwd[2 * a + 1] = (int16_t)c.bw_v;
                s->mb_dir[a] = (uint8_t)c.prev_dir;
            }
            increment--;
        }
        c.mb_addr++;
    }
    const int mb = c.mb_addr;
    if (mb < 0 || mb >= s->mbsize) { c.err = "macroblock address outside the picture"; return 0; }
    // row and column follow the address without a division (addresses only grow inside a slice)
    c.mb_col += mb - c.rc_addr;
    c.rc_addr = mb;
    while (c.mb_col >= s->mbw) { c.mb_col -= s->mbw; c.mb_row++; }
    if (!c.run || c.run->mb_row != c.mb_row) {
        SliceOut& so = *c.sout;
        if (so.used == so.runs.size()) so.runs.emplace_back();
        c.run = &so.runs[so.used++];
        c.run->mb_row = c.mb_row;
        for (int k = 0; k < s->n_streams; k++) {
            c.run->used[k] = 0;
            c.run->cnt[k].assign((size_t)(k == 2 || k == 3 ? s->gc : s->gy), 0u);
        }
    }
    const int mb_type = r.vlc(T.mbtype[type]);
    if (r.bad) { c.err = "invalid macroblock type"; return 0; }
    c.mb_intra = mb_type & 0x01;
    c.mot_fw = mb_type & 0x08;
    c.mot_bw = mb_type & 0x04;
    if (mb_type & 0x10) c.qs = (int)r.get(5);
    s->qscale[mb] = (uint8_t)c.qs;
    s->intra[mb] = c.mb_intra ? 255 : 0;
    if (c.mb_intra) {
        c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0;
        c.bw_h = c.bw_h_prev = 0; c.bw_v = c.bw_v_prev = 0;
        c.prev_dir = 0;
        if (type != 1) s->repadd[mb] = 255;                        // jsv.js:1502-1505
    } else {
        c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
        decode_motion_vectors(c);
        if (r.bad) { c.err = "invalid motion code"; return 0; }
        s->mv_fwd[2 * mb] = (int16_t)c.fw_h;
        s->mv_fwd[2 * mb + 1] = (int16_t)c.fw_v;
        if (type == 3) {
            s->mv_bwd[2 * mb] = (int16_t)c.bw_h;
            s->mv_bwd[2 * mb + 1] = (int16_t)c.bw_v;
            c.prev_dir = (c.mot_fw ? 1 : 0) | (c.mot_bw ? 2 : 0);
            s->mb_dir[mb] = (uint8_t)c.prev_dir;
        }
    }
    int cbp = 0;
    if (mb_type & 0x02) { cbp = r.vlc(T.cbp); if (r.bad) { c.err = "invalid coded block pattern"; return 0; } }
    else if (c.mb_intra) cbp = 0x3f;
    // yuva (the repo's syntax, tools/jsv_writer.py write_picture): four A blocks after Cr -- all of them in an intra
    // macroblock, else those of a 4-bit alpha_pattern that every non-intra macroblock carries here
    int apat = 0;
    if (s->n_streams == 6) apat = c.mb_intra ? 0xf : (int)r.get(4);
    for (int block = 0, mask = 0x20; block < 6; block++, mask >>= 1)
        if (cbp & mask) { if (!decode_block(c, block)) return 0; }
    for (int block = 6, mask = 0x8; block < 10; block++, mask >>= 1)
        if (apat & mask) { if (!decode_block(c, block)) return 0; }
    return 1;
}

// decoders/jsv.js:683-706
void decode_slice(leon_vlc_stream* s, const SliceJob& job, SliceOut* out)
{
    SliceCtx c;
    c.s = s;
    c.r.b = s->r.b;
    c.r.nbytes = s->r.nbytes;
    c.r.pos = job.bitpos;
    c.sout = out;
    out->used = 0;
    c.mb_addr = (job.code - 1) * s->mbw - 1;
    c.mb_row = job.code - 1;
    c.mb_col = -1;
    c.rc_addr = c.mb_addr;
    c.qs = (int)c.r.get(5);
    while (c.r.get(1) && !c.r.bad) c.r.get(8);
    do {
        if (decode_macroblock(c) == 0) break;
    } while (!c.r.next_bits_are_start_code() && !c.r.bad);
    if (c.r.bad && !c.err) c.err = "bitstream ends inside a slice";
    if (c.err) {
        std::lock_guard<std::mutex> lk(s->err_mu);
        if (!s->slice_error.exchange(1))
            snprintf(s->slice_err_text, sizeof(s->slice_err_text), "slice %d: %s near byte %zu", job.code, c.err, c.r.pos >> 3);
    }
}

void run_jobs(leon_vlc_stream* s, int tid)
{
    for (;;) {
        const size_t j = s->next_job.fetch_add(1);
        if (j >= s->jobs.size()) break;
        decode_slice(s, s->jobs[j], &s->slice_out[j]);
        (void)tid;
    }
}

// Workers spin briefly before they sleep: the slices of a 1080p picture are ~30 us of work
// each, and a condition-variable wake-up costs about as much as two of them.
void worker_main(leon_vlc_stream* s, int tid)
{
    uint64_t seen = 0;
    for (;;) {
        int spins = 0;
        while (s->generation.load(std::memory_order_acquire) == seen && !s->quit.load(std::memory_order_relaxed)) {
            if (++spins < 4000) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                continue;
            }
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_work.wait(lk, [&] { return s->quit.load() || s->generation.load() != seen; });
        }
        if (s->quit.load()) return;
        seen = s->generation.load(std::memory_order_acquire);
        run_jobs(s, tid);
        if (s->busy.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(s->mu);
            s->cv_done.notify_one();
        }
    }
}

// ---- headers ------------------------------------------------------------------------------------

void init_buffers(leon_vlc_stream* s)        // decoders/jsv.js:355-423
{
    leon_vlc_info& I = s->info;
    I.mb_width = (I.frame_width + 15) >> 4;
    I.mb_height = (I.frame_height + 15) >> 4;
    I.coded_width = I.mb_width << 4;
    I.coded_height = I.mb_height << 4;
    s->mbw = I.mb_width;
    s->mbh = I.mb_height;
    s->mbsize = s->mbw * s->mbh;
    s->gy = (2 * s->mbw + 7) >> 3;
    s->gc = (s->mbw + 7) >> 3;
    s->n_y = 2 * s->mbh * s->gy;
    s->n_c = s->mbh * s->gc;
    I.groups_y = s->gy;
    I.groups_c = s->gc;
    s->n_streams = I.has_alpha == 1 ? 6 : 4;
    I.n_groups = s->n_y + 2 * s->n_c + (s->n_streams == 6 ? s->n_y : 0);
    s->qscale.assign((size_t)s->mbsize, 0);
    s->intra.assign((size_t)s->mbsize, 0);
    s->sequence_started = true;
}

bool decode_sequence_header(leon_vlc_stream* s)      // decoders/jsv.js:491-561
{
    Bits& r = s->r;
    leon_vlc_info& I = s->info;
    const int fw = (int)r.get(12), fh = (int)r.get(12);
    r.skip(4);
    const double rate = kPictureRate[r.get(4)];
    r.skip(18 + 1);
    r.get(10);
    r.skip(1);
    uint8_t intra[64], non[64];
    memcpy(intra, kDefaultIntra, 64);
    memset(non, 16, 64);
    if (r.get(1)) for (int i = 0; i < 64; i++) intra[kZigZag[i]] = (uint8_t)r.get(8);
    if (r.get(1)) for (int i = 0; i < 64; i++) non[kZigZag[i]] = (uint8_t)r.get(8);
    if (r.bad || fw <= 0 || fh <= 0) return false;
    if (!s->sequence_started) {
        I.frame_width = fw;
        I.frame_height = fh;
        I.picture_rate = rate;
        init_buffers(s);
    }
    memcpy(I.intra_qm, intra, 64);
    memcpy(I.non_intra_qm, non, 64);
    s->new_sequence = true;
    return true;
}

void decode_gop_header(leon_vlc_stream* s)           // decoders/jsv.js:471-489
{
    Bits& r = s->r;
    r.skip(1);
    const int h = (int)r.get(5), m = (int)r.get(6);
    r.skip(1);
    const int sec = (int)r.get(6), f = (int)r.get(6);
    const double rate = s->info.picture_rate > 0 ? s->info.picture_rate : 25.0;
    s->ts_pending = ((h * 60 + m) * 60 + sec + (f + 1) / rate) * 1000.0;
    s->open_gop_pending = r.get(1) == 0;                  // closed_gop; broken_link follows
}

// decoders/jsv.js:583-676 (+ B pictures).  1 = picture decoded, 0 = not a picture we read, <0 error
int decode_picture(leon_vlc_stream* s, leon_vlc_picture* out)
{
    Bits& r = s->r;
    s->temporal_reference = (int)r.get(10);
    const int type = (int)r.get(3);
    r.skip(16);
    if (type <= 0 || type > 3) return 0;
    s->type = type;
    s->mv_fwd.assign((size_t)s->mbsize * 2, 0);
    if (type != 1) {
        s->repadd.assign((size_t)s->mbsize, 0);
        s->full_pel_fwd = (int)r.get(1);
        const int fcode = (int)r.get(3);
        if (fcode == 0) return 0;
        s->fwd_rsize = fcode - 1;
        s->fwd_f = 1 << s->fwd_rsize;
    }
    if (type == 3) {
        s->mv_bwd.assign((size_t)s->mbsize * 2, 0);
        s->mb_dir.assign((size_t)s->mbsize, 0);
        s->full_pel_bwd = (int)r.get(1);
        const int bcode = (int)r.get(3);
        if (bcode == 0) return 0;
        s->bwd_rsize = bcode - 1;
        s->bwd_f = 1 << s->bwd_rsize;
    }
    // locate the slices; the picture ends at the first start code that is not a slice
    int code;
    do { code = r.next_start_code(); } while (code == START_EXTENSION || code == START_USER_DATA);
    s->jobs.clear();
    while (code >= START_SLICE_FIRST && code <= START_SLICE_LAST) {
        s->jobs.push_back(SliceJob{code, r.pos});
        code = r.next_start_code();
    }
    if (code >= 0) r.pos -= 32;                                    // rewind(32)
    if (s->scan_only) {                                            // leon_vlc_scan_picture: the slices are not decoded here
        s->scan_code.clear();
        s->scan_pos.clear();
        for (const SliceJob& j : s->jobs) { s->scan_code.push_back(j.code); s->scan_pos.push_back((uint64_t)j.bitpos); }
        s->scan_end = (uint64_t)(r.pos >> 3);
        out->type = type;
        out->temporal_reference = s->temporal_reference;
        out->ts_ms = s->ts_pending;
        s->ts_pending = 0;
        out->new_sequence = s->new_sequence ? 1 : 0;
        s->new_sequence = false;
        out->open_gop = s->open_gop_pending ? 1 : 0;
        s->open_gop_pending = false;
        out->n_groups = s->info.n_groups;
        out->n_slices = (uint32_t)s->jobs.size();
        return 1;
    }

    if (s->slice_out.size() < s->jobs.size()) s->slice_out.resize(s->jobs.size());
    for (size_t j = 0; j < s->jobs.size(); j++) s->slice_out[j].used = 0;
    s->next_job.store(0);
    s->slice_error.store(0);
    const int helpers = (int)s->workers.size();
    if (helpers > 0 && s->jobs.size() > 1) {
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->busy.store(helpers);
            s->generation.fetch_add(1, std::memory_order_release);
        }
        s->cv_work.notify_all();
        run_jobs(s, 0);
        for (int spins = 0; s->busy.load(std::memory_order_acquire) != 0 && spins < 4000; spins++) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (s->busy.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_done.wait(lk, [&] { return s->busy.load() == 0; });
        }
    } else {
        run_jobs(s, 0);
    }
    if (s->slice_error.load()) return fail(LEON_VLC_ERR_STREAM, "%s", s->slice_err_text);

    // assemble: per-group counts from every run, prefix sums, then the run streams are copied group
    // by group (a group that two slices share -- a slice boundary inside it -- stays contiguous)
    const size_t ng = (size_t)s->info.n_groups;
    s->grp_off.assign(ng + 1, 0);
    auto base_of = [&](const RowRun& rr, int k) -> size_t {
        if (k >= 4) return (size_t)s->n_y + 2 * (size_t)s->n_c + (size_t)(2 * rr.mb_row + k - 4) * (size_t)s->gy;
        return k < 2 ? (size_t)(2 * rr.mb_row + k) * (size_t)s->gy
                     : (size_t)s->n_y + (k == 3 ? (size_t)s->n_c : 0) + (size_t)rr.mb_row * (size_t)s->gc;
    };
    for (size_t j = 0; j < s->jobs.size(); j++)
        for (size_t u = 0; u < s->slice_out[j].used; u++) {
            const RowRun& rr = s->slice_out[j].runs[u];
            for (int k = 0; k < s->n_streams; k++) {
                const size_t base = base_of(rr, k);
                for (size_t g = 0; g < rr.cnt[k].size(); g++) s->grp_off[base + g + 1] += rr.cnt[k][g];
            }
        }
    for (size_t g = 0; g < ng; g++) s->grp_off[g + 1] += s->grp_off[g];
    s->entries.resize(s->grp_off[ng]);
    s->cursor.assign(s->grp_off.begin(), s->grp_off.end() - 1);
    for (size_t j = 0; j < s->jobs.size(); j++)
        for (size_t u = 0; u < s->slice_out[j].used; u++) {
            const RowRun& rr = s->slice_out[j].runs[u];
            for (int k = 0; k < s->n_streams; k++) {
                const size_t base = base_of(rr, k);
                const uint32_t* src = rr.ent[k].data();
                for (size_t g = 0; g < rr.cnt[k].size(); g++) {
                    const uint32_t n = rr.cnt[k][g];
                    if (!n) continue;
                    memcpy(s->entries.data() + s->cursor[base + g], src, (size_t)n * 4);
                    s->cursor[base + g] += n;
                    src += n;
                }
            }
        }

    out->type = type;
    out->temporal_reference = s->temporal_reference;
    out->ts_ms = s->ts_pending;
    s->ts_pending = 0;
    out->new_sequence = s->new_sequence ? 1 : 0;
    s->new_sequence = false;
    out->open_gop = s->open_gop_pending ? 1 : 0;
    s->open_gop_pending = false;
    out->n_groups = s->info.n_groups;
    out->n_entries = (uint32_t)s->entries.size();
    out->grp_off = s->grp_off.data();
    out->entries = s->entries.data();
    out->qscale = s->qscale.data();
    out->intra = s->intra.data();
    out->repadd = type != 1 ? s->repadd.data() : nullptr;
    out->mv_fwd = type != 1 ? s->mv_fwd.data() : nullptr;
    out->mv_bwd = type == 3 ? s->mv_bwd.data() : nullptr;
    out->mb_dir = type == 3 ? s->mb_dir.data() : nullptr;
    out->n_slices = (uint32_t)s->jobs.size();
    return 1;
}

}  // namespace

namespace {
void ahead_main(leon_vlc_stream* s);
void ahead_wait(leon_vlc_stream* s);
}  // namespace

extern "C" {

int leon_vlc_abi_version(void) { return LEON_VLC_ABI_VERSION; }

const char* leon_vlc_last_error(void) { return g_err; }

int leon_vlc_open(const uint8_t* data, size_t n, int32_t threads, leon_vlc_stream** out)
{
    return leon_vlc_open_shard(data, n, threads, -1, out);
}

// borrow: the caller's bytes are read in place (16 readable bytes behind them, any content) and only the layers above
// the slices will be asked for -- no copy, no read-ahead thread
static int open_stream(const uint8_t* data, size_t n, int32_t threads, int32_t has_alpha, bool borrow, leon_vlc_stream** out)
{
    if (!data || !out || n < 12) return fail(LEON_VLC_ERR_INVALID, "null or too short stream");
    leon_vlc_stream* s = new (std::nothrow) leon_vlc_stream();
    if (!s) return fail(LEON_VLC_ERR_NOMEM, "out of memory");
    if (borrow) s->r.b = data;
    else {
        s->data.assign(data, data + n);
        s->data.resize(n + 16, 0);
        s->r.b = s->data.data();
    }
    s->r.nbytes = n;
    s->r.pos = 0;
    (void)tables();
    Bits& r = s->r;
    s->raw_es = data[0] == 0 && data[1] == 0 && data[2] == 1 && data[3] == START_SEQUENCE_ES;
    // a GOP shard cut out of a JSV stream at a key-map entry starts with the JSV sequence header itself
    const bool shard = data[0] == 0 && data[1] == 0 && data[2] == 1 && data[3] == START_SEQUENCE;
    s->info.has_alpha = shard && has_alpha == 1 ? 1 : -1;      // the flag lives in the container header a shard lacks
    if (!s->raw_es && !shard) {
        // container header: decoders/jsv.js:237-313
        r.skip(16);
        r.get(16);
        r.get(16);
        double d = r.get(16) / 100.0;
        s->info.has_alpha = -1;
        if (d == 0) { s->info.has_alpha = (int)r.get(1); d = r.get(23) / 100.0; }
        s->info.duration = d;
        const size_t i = r.pos >> 3;
        if (i + 12 <= n && r.b[i] == 0 && r.b[i + 1] == 0 && r.b[i + 2] == 1 && r.b[i + 3] == START_MAP) {
            r.skip(32);
            const uint32_t count = r.get(32);
            if ((size_t)count * 8 > n) { delete s; return fail(LEON_VLC_ERR_STREAM, "key map larger than the stream"); }
            s->keymap.resize((size_t)count * 2);
            for (uint32_t k = 0; k < count; k++) { s->keymap[2 * k] = r.get(32); s->keymap[2 * k + 1] = r.get(32); }
            s->info.keymap_count = count;
        }
    }
    s->have_meta = true;
    // read ahead to the first sequence header so that the geometry is known
    const size_t resume = r.pos;
    bool found = false;
    for (;;) {
        const int code = r.next_start_code();
        if (code < 0) break;
        if (code == START_SEQUENCE || (s->raw_es && code == START_SEQUENCE_ES)) {
            if (!decode_sequence_header(s)) { delete s; return fail(LEON_VLC_ERR_STREAM, "bad sequence header"); }
            found = true;
            break;
        }
    }
    if (!found) { delete s; return fail(LEON_VLC_ERR_STREAM, "no sequence header in the stream"); }
    r.pos = resume;
    s->new_sequence = false;

    int nt = threads;
    if (nt <= 0) { nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16; }
    if (nt > 64) nt = 64;
    s->n_threads = nt;
    s->info.threads = (uint32_t)nt;
    for (int t = 1; t < nt; t++) s->workers.emplace_back(worker_main, s, t);
    s->info_out = s->info;
    s->scan_stream = borrow;
    if (!borrow) s->ahead = std::thread(ahead_main, s);
    *out = s;
    return LEON_VLC_OK;
}

int leon_vlc_open_shard(const uint8_t* data, size_t n, int32_t threads, int32_t has_alpha, leon_vlc_stream** out)
{
    return open_stream(data, n, threads, has_alpha, false, out);
}

int leon_vlc_open_scan(const uint8_t* data, size_t n, size_t readable, int32_t has_alpha, leon_vlc_stream** out)
{
    if (readable < n) return fail(LEON_VLC_ERR_INVALID, "readable bytes fewer than the stream's");
    return open_stream(data, n, 1, has_alpha, readable >= n + 16, out);
}

void leon_vlc_close(leon_vlc_stream* s)
{
    if (!s) return;
    if (s->a_inflight >= 0) ahead_wait(s);
    {
        std::lock_guard<std::mutex> lk(s->amu);
        s->a_quit = true;
    }
    s->acv.notify_all();
    if (s->ahead.joinable()) s->ahead.join();
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->quit.store(true);
    }
    s->cv_work.notify_all();
    for (auto& t : s->workers) t.join();
    delete s;
}

int leon_vlc_get_info(leon_vlc_stream* s, leon_vlc_info* out)
{
    if (!s || !out) return fail(LEON_VLC_ERR_INVALID, "null argument");
    *out = s->info_out;
    return LEON_VLC_OK;
}

}  // extern "C"

namespace {

// = decodeFrame: parse up to and including the next picture into the stream's working arrays
int next_picture_sync(leon_vlc_stream* s, leon_vlc_picture* out)
{
    if (s->ended) return LEON_VLC_END;
    Bits& r = s->r;
    for (;;) {                                                      // decoders/jsv.js:426-469
        const int code = r.next_start_code();
        if (code < 0) { s->ended = true; return LEON_VLC_END; }
        if (code == START_SEQUENCE || (s->raw_es && code == START_SEQUENCE_ES)) {
            if (!decode_sequence_header(s)) return fail(LEON_VLC_ERR_STREAM, "bad sequence header near byte %zu", r.pos >> 3);
            s->skip_till_gop = false;
            continue;
        }
        if (s->skip_till_gop) continue;
        if (code == START_GOP) { decode_gop_header(s); continue; }
        if (code == START_PICTURE && s->sequence_started) {
            const int rc = decode_picture(s, out);
            if (rc != 0) return rc;
        }
    }
}

// parse the next picture and move it into result set k (vector storage is swapped, not copied;
// the two persistent maps are copied: the next picture keeps updating them)
void fill_result(leon_vlc_stream* s, int k)
{
    leon_vlc_stream::Result& R = s->results[k];
    leon_vlc_picture p{};
    g_err[0] = 0;
    R.rc = next_picture_sync(s, &p);
    snprintf(R.err, sizeof(R.err), "%s", g_err);
    R.info = s->info;
    if (R.rc == LEON_VLC_PICTURE) {
        R.grp_off.swap(s->grp_off);
        R.entries.swap(s->entries);
        R.repadd.swap(s->repadd);
        R.mv_fwd.swap(s->mv_fwd);
        R.mv_bwd.swap(s->mv_bwd);
        R.mb_dir.swap(s->mb_dir);
        R.qscale = s->qscale;
        R.intra = s->intra;
        p.grp_off = R.grp_off.data();
        p.entries = R.entries.data();
        p.qscale = R.qscale.data();
        p.intra = R.intra.data();
        p.repadd = p.type != 1 ? R.repadd.data() : nullptr;
        p.mv_fwd = p.type != 1 ? R.mv_fwd.data() : nullptr;
        p.mv_bwd = p.type == 3 ? R.mv_bwd.data() : nullptr;
        p.mb_dir = p.type == 3 ? R.mb_dir.data() : nullptr;
    }
    R.pic = p;
}

void ahead_main(leon_vlc_stream* s)
{
    for (;;) {
        int k;
        {
            std::unique_lock<std::mutex> lk(s->amu);
            s->acv.wait(lk, [&] { return s->a_quit || s->a_request >= 0; });
            if (s->a_quit) return;
            k = s->a_request;
            s->a_request = -1;
        }
        fill_result(s, k);
        {
            std::lock_guard<std::mutex> lk(s->amu);
            s->a_done = true;
        }
        s->acv.notify_all();
    }
}

void ahead_start(leon_vlc_stream* s, int k)         // caller holds no lock
{
    {
        std::lock_guard<std::mutex> lk(s->amu);
        s->a_request = k;
        s->a_inflight = k;
        s->a_done = false;
    }
    s->acv.notify_all();
}

void ahead_wait(leon_vlc_stream* s)
{
    std::unique_lock<std::mutex> lk(s->amu);
    s->acv.wait(lk, [&] { return s->a_done; });
}

}  // namespace

extern "C" {

int leon_vlc_next_picture(leon_vlc_stream* s, leon_vlc_picture* out)
{
    if (!s || !out) return fail(LEON_VLC_ERR_INVALID, "null argument");
    if (s->scan_stream) return fail(LEON_VLC_ERR_INVALID, "a stream of leon_vlc_open_scan serves leon_vlc_scan_picture only");
    if (s->a_eos) return LEON_VLC_END;
    if (s->a_inflight < 0) ahead_start(s, 0);
    ahead_wait(s);
    const int k = s->a_inflight;
    leon_vlc_stream::Result& R = s->results[k];
    s->info_out = R.info;
    if (R.rc == LEON_VLC_END) {
        s->a_eos = true;
        s->a_inflight = -1;
        return LEON_VLC_END;
    }
    ahead_start(s, 1 - k);                          // the other set: `R` stays untouched until the next call
    if (R.rc < 0) return fail(R.rc, "%s", R.err);
    *out = R.pic;
    return LEON_VLC_PICTURE;
}

int leon_vlc_next_picture_sync(leon_vlc_stream* s, leon_vlc_picture* out)
{
    if (!s || !out) return fail(LEON_VLC_ERR_INVALID, "null argument");
    if (s->scan_stream) return fail(LEON_VLC_ERR_INVALID, "a stream of leon_vlc_open_scan serves leon_vlc_scan_picture only");
    if (s->a_inflight >= 0 || s->a_eos) return fail(LEON_VLC_ERR_INVALID, "leon_vlc_next_picture has been used on this stream: do not mix the two");
    g_err[0] = 0;
    const int rc = next_picture_sync(s, out);
    s->info_out = s->info;
    return rc;
}

int leon_vlc_scan_picture(leon_vlc_stream* s, leon_vlc_picture_scan* out)
{
    if (!s || !out) return fail(LEON_VLC_ERR_INVALID, "null argument");
    if (s->ended) return LEON_VLC_END;
    g_err[0] = 0;
    leon_vlc_picture p{};
    s->scan_only = true;
    const int rc = next_picture_sync(s, &p);
    s->scan_only = false;
    s->info_out = s->info;
    if (rc != LEON_VLC_PICTURE) return rc;
    memset(out, 0, sizeof(*out));
    out->type = p.type;
    out->temporal_reference = p.temporal_reference;
    out->ts_ms = p.ts_ms;
    out->new_sequence = p.new_sequence;
    out->open_gop = p.open_gop;
    out->full_pel_fwd = s->full_pel_fwd; out->fwd_rsize = s->fwd_rsize;
    out->full_pel_bwd = s->full_pel_bwd; out->bwd_rsize = s->bwd_rsize;
    out->n_slices = p.n_slices;
    out->slice_code = s->scan_code.data();
    out->slice_bit_pos = s->scan_pos.data();
    out->end_byte = s->scan_end;
    return LEON_VLC_PICTURE;
}

int leon_vlc_get_gpu_tables(leon_vlc_gpu_tables* out)
{
    if (!out) return fail(LEON_VLC_ERR_INVALID, "null argument");
    const Tables& T = tables();
    memset(out, 0, sizeof(*out));
    memcpy(out->fast12, T.fast12, sizeof(out->fast12));
    memcpy(out->motion_s, T.motion_s, sizeof(out->motion_s));
    for (size_t i = 0; i < 65536; i++) out->coef16[i] = T.coef.t[i];
    for (size_t i = 0; i < 2048; i++) out->mba[i] = T.mba.t[i];
    for (int t = 1; t <= 3; t++)
        for (size_t i = 0; i < 64; i++) out->mbtype[t][i] = T.mbtype[t].t[i >> (6 - T.mbtype[t].max_len)];
    for (size_t i = 0; i < 512; i++) out->cbp[i] = T.cbp.t[i];
    for (size_t i = 0; i < 128; i++) out->dc_lum[i] = T.dc_lum.t[i];
    for (size_t i = 0; i < 256; i++) out->dc_chr[i] = T.dc_chr.t[i];
    for (int i = 0; i < 64; i++) out->zz_off[i] = (uint16_t)((kZigZag[i] >> 3) * 128 + (kZigZag[i] & 7) * 2);
    return LEON_VLC_OK;
}

int leon_vlc_seek(leon_vlc_stream* s, double seconds, uint64_t* byte_offset)
{
    if (!s) return fail(LEON_VLC_ERR_INVALID, "null argument");
    uint64_t offset = 0;
    const uint32_t count = s->info_out.keymap_count;
    if (count) {
        const double rate = s->info_out.picture_rate > 0 ? s->info_out.picture_rate : 25.0;
        auto key_time = [&](uint32_t g) {                           // decoders/jsv.js:315-325
            const uint32_t tc = s->keymap[2 * g + 1];
            const int hour = (tc >> 26) & 31, minute = (tc >> 20) & 63, second = (tc >> 13) & 63, frame = (tc >> 7) & 63;
            return (hour * 60 + minute) * 60 + second + (frame + 1) / rate;
        };
        const double dur = s->info_out.duration > 0 ? s->info_out.duration : 1.0;
        double gf = (double)count * seconds / dur;
        if (gf < 0) gf = 0;
        uint32_t g = (uint32_t)gf;
        if (g > count - 1) g = count - 1;
        while (g > 0 && key_time(g) > seconds + 1e-9) g--;
        while (g + 1 < count && key_time(g + 1) <= seconds + 1e-9) g++;
        offset = s->keymap[2 * g];
    }
    if (offset > s->r.nbytes) return fail(LEON_VLC_ERR_STREAM, "key map entry beyond the stream");
    if (s->a_inflight >= 0) ahead_wait(s);          // whatever was parsed ahead is dropped
    s->a_inflight = -1;
    s->a_eos = false;
    s->r.pos = (size_t)offset * 8;
    s->ended = false;
    s->skip_till_gop = true;
    if (byte_offset) *byte_offset = offset;
    return LEON_VLC_OK;
}

int leon_vlc_get_keymap(leon_vlc_stream* s, uint32_t* byte_offsets, uint32_t* timecodes, uint32_t capacity)
{
    if (!s) return fail(LEON_VLC_ERR_INVALID, "null argument");
    const uint32_t count = (uint32_t)(s->keymap.size() / 2);
    for (uint32_t g = 0; g < count && g < capacity; g++) {
        if (byte_offsets) byte_offsets[g] = s->keymap[2 * g];
        if (timecodes) timecodes[g] = s->keymap[2 * g + 1];
    }
    return (int)count;
}

int leon_vlc_densify(const leon_vlc_info* I, const leon_vlc_picture* p, int16_t* y, int16_t* cb, int16_t* cr)
{
    if (!I || !p || !y || !cb || !cr) return fail(LEON_VLC_ERR_INVALID, "null argument");
    const int cw = I->coded_width, ch = I->coded_height, hw = cw >> 1;
    memset(y, 0, sizeof(int16_t) * (size_t)cw * ch);
    memset(cb, 0, sizeof(int16_t) * (size_t)(cw >> 1) * (ch >> 1));
    memset(cr, 0, sizeof(int16_t) * (size_t)(cw >> 1) * (ch >> 1));
    const int n_y = 2 * I->mb_height * I->groups_y, n_c = I->mb_height * I->groups_c;
    for (int g = 0; g < p->n_groups && g < n_y + 2 * n_c; g++) {      // a yuva picture's A groups: leon_vlc_densify_alpha
        int16_t* plane;
        int stride, R, gg, bw;
        if (g < n_y) { plane = y; stride = cw; R = g / I->groups_y; gg = g % I->groups_y; bw = cw >> 3; }
        else {
            const int k = (g - n_y) % n_c;
            plane = g - n_y < n_c ? cb : cr;
            stride = hw; R = k / I->groups_c; gg = k % I->groups_c; bw = hw >> 3;
        }
        for (uint32_t e = p->grp_off[g]; e < p->grp_off[g + 1]; e++) {
            const uint32_t v = p->entries[e], off = (v >> 16) & 1023u;
            const int r = (int)(off >> 7), b = (int)((off >> 4) & 7), c = (int)((off >> 1) & 7);
            const int q = gg * 8 + b;
            if (q >= bw) return fail(LEON_VLC_ERR_INVALID, "entry outside the plane");
            plane[(size_t)(R * 8 + r) * stride + q * 8 + c] = (int16_t)(v & 0xffffu);
        }
    }
    return LEON_VLC_OK;
}

int leon_vlc_densify_alpha(const leon_vlc_info* I, const leon_vlc_picture* p, int16_t* a)
{
    if (!I || !p || !a) return fail(LEON_VLC_ERR_INVALID, "null argument");
    const int cw = I->coded_width, ch = I->coded_height;
    const int n_y = 2 * I->mb_height * I->groups_y, n_c = I->mb_height * I->groups_c;
    if (p->n_groups != 2 * n_y + 2 * n_c) return fail(LEON_VLC_ERR_INVALID, "not a yuva picture");
    memset(a, 0, sizeof(int16_t) * (size_t)cw * ch);
    for (int g = n_y + 2 * n_c; g < p->n_groups; g++) {
        const int k = g - n_y - 2 * n_c, R = k / I->groups_y, gg = k % I->groups_y;
        for (uint32_t e = p->grp_off[g]; e < p->grp_off[g + 1]; e++) {
            const uint32_t v = p->entries[e], off = (v >> 16) & 1023u;
            const int r = (int)(off >> 7), b = (int)((off >> 4) & 7), c = (int)((off >> 1) & 7);
            const int q = gg * 8 + b;
            if (q >= (cw >> 3)) return fail(LEON_VLC_ERR_INVALID, "entry outside the plane");
            a[(size_t)(R * 8 + r) * cw + q * 8 + c] = (int16_t)(v & 0xffffu);
        }
    }
    return LEON_VLC_OK;
}

}  // extern "C"
