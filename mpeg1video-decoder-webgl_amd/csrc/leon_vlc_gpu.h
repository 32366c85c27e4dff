// leon_vlc_gpu.h -- the slice layer of the bitstream front end on the GPU (gfx950), for the pipeline's
// gpu_parser mode (include/leon_pipeline.h).
//
// What it replaces: decodeSlice -> decodeMacroblock -> decodeMotionVectors / decodeBlockGL of the reference
// (decoders/jsv.js:683-706, :725-828, :831-893, :1338-1525), i.e. everything below a slice start code, exactly as
// csrc/leon_vlc.cpp decodes it on the host (same tables -- leon_vlc_get_gpu_tables --, same decisions, same
// error conditions; tests/test_gpu_parser_gpu.py holds the two against each other tensor by tensor).  The layers
// above -- container, sequence / GOP / picture headers, the slice start codes -- stay on the host
// (leon_vlc_scan_picture): they are a few dozen bytes per picture.
//
// Why here: a slice is sequential, but a 1080p picture has 68 of them and a window of the pipeline holds
// 128 GOPs x 12 pictures: ~100 000 independent slices, one LANE each.  16 host cores parse 14 k pictures/s;
// the reconstruction kernels take 220 k/s.
//
//   k_vlc_parse    one lane per slice: macroblock maps straight into the picture's arrays; coefficients into the
//                  slice's scratch strip -- one header word {group id, count} per coded block at its front, the
//                  entries behind, both in decoding order -- and the count added to the group's counter.
//   k_vlc_offsets  one workgroup per picture: exclusive scan of the group counters -> grp_off (what
//                  leon_sparse_picture wants), counters back to zero (they become cursors).
//   k_vlc_gather   one WAVE per slice: 64 block headers at a time, a wave prefix sum of their counts finds each
//                  block's entries, which move to grp_off[group] + cursor (atomic add of the block's count).
//                  Entries of a group are "in no particular order" (include/leon_vlc.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace leon {

// Pointers into device memory are told to be GLOBAL: through a generic pointer every store is a flat_store, which
// counts on the LDS counter too -- and the wait in front of the next table lookup (LDS) then waits for the entry
// stored a symbol earlier to reach memory.
#define VLC_G __attribute__((address_space(1)))

// Device copy of the front end's tables (leon_vlc_get_gpu_tables), the LDS part first and in 16 bits:
//   fast12: bits 0..3 length (0 = longer code or escape), bit 4 end of block, bits 5..9 run, bits 10..15 level
//   long9:  the codes of 12 .. 16 bits (all of them start with seven zeros) by the nine bits behind those zeros:
//           bits 0..4 length without the sign bit, bits 5..9 run, bits 10..15 level (unsigned), 0 = invalid code
//   the others: (length << 8) | value, 0 = invalid code
struct VlcTables {
    uint16_t fast12[4096];
    uint16_t long9[512];
    uint16_t motion_s[2048];
    uint16_t mba[2048];
    uint16_t cbp[512];
    uint16_t mbtype[4][64];
    uint16_t dc_lum[128], dc_chr[256];
    uint16_t zz_off[64];
};
static constexpr int kVlcLdsWords = (4096 + 512 + 2048 + 2048 + 512 + 256 + 128 + 256 + 64) / 2;
// every lane reads its slice through a ring of 16 dwords in LDS: dword i of lane L at ring[(i & 15) * 64 + L]
static constexpr int kVlcRingDwords = 16;
static constexpr int kVlcRingBytesPerWave = kVlcRingDwords * 64 * 4;

struct VlcGeom {
    int32_t mbw, mbh, gy, gc, n_y, n_c, n_groups, alpha;
    // byte offsets of a picture's arrays from VlcPic::zbase (the same for every picture: wave-uniform)
    uint32_t off_cnt, off_qscale, off_intra, off_repadd, off_mb_dir, off_mv_fwd, off_mv_bwd, pad;
};

struct VlcPic {                  // one picture of the window
    int32_t type, full_pel_fwd, fwd_rsize, full_pel_bwd, bwd_rsize, pad;
    char* zbase;                 // zero on entry: group counters [n_groups + 1] and the macroblock maps, at VlcGeom's offsets
    uint32_t* grp_off;           // [n_groups + 1]
    uint32_t* entries;
    uint32_t entries_cap;
    uint32_t pad2;
};

struct VlcSlice {
    const uint32_t* bytes;       // the GOP's stream copy (4-byte aligned, zero padded)
    uint32_t n_dwords;           // readable dwords of it
    uint32_t n_bytes;            // its real length
    uint32_t bit_pos;            // first bit behind the slice start code
    uint32_t end_byte;           // first byte behind the slice
    int32_t code;                // slice_vertical_position
    uint32_t pic;
    uint32_t hdr_cap;            // block headers the strip has room for ...
    uint32_t ent_cap;            // ... and entries behind them
    uint32_t* scratch;           // [hdr_cap headers][ent_cap entries]
};

enum { VLC_ERR_MBA = 1, VLC_ERR_ADDR, VLC_ERR_TYPE, VLC_ERR_MOTION, VLC_ERR_CBP, VLC_ERR_COEF, VLC_ERR_INDEX, VLC_ERR_END, VLC_ERR_DC,
       VLC_ERR_SCRATCH };

// Requests dwords [loaded, upto) of a lane's stream into its ring slots: 16 conditional LDS-direct loads, one per
// slot (the LDS address of such a load is wave-uniform: slot i of every lane that wants it goes in one instruction).
// NOT inlined: the top-up sits in front of every syntax element (28 places), and 28 copies of this made the kernel
// four times as long as the instruction cache.
__device__ __attribute__((noinline)) void vlc_request(const uint32_t* base, uint32_t loaded, uint32_t upto, uint32_t nd, uint32_t* wave_ring, int lane)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t idx = loaded + (((uint32_t)i - loaded) & 15u);      // the dword >= loaded that lives in slot i
        if (idx < upto) {
            if (idx < nd) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + idx),
                                                            (__attribute__((address_space(3))) void*)(wave_ring + i * 64), 4, 0, 0);
            else wave_ring[i * 64 + lane] = 0u;
        }
    }
}

// The lane's view of its slice.  The stream does not come from memory symbol by symbol -- with 64 lanes in a wave some
// lane would be waiting for a load at every step, and the wave with it -- but through the ring: the lanes top their
// rings up TOGETHER (sync, below) with loads that go straight to LDS, half a ring ahead of what is being read, so
// that the wait in front of the next top-up finds them long landed.  Reading a dword is an LDS access.
struct VlcWin {
    const uint32_t* base;
    uint32_t next, nd;           // next dword to take from the ring, dwords the stream copy has
    uint32_t safe, loaded;       // [next, safe) is in the ring, [safe, loaded) is on its way
    uint64_t w;                  // the stream from `pos` on, left aligned, `avail` bits valid, zeros below
    int avail;
    uint32_t pos;                // in bits, from base
    uint32_t ahead;              // dword `next` of the stream as it lies in memory (big endian), already out of the ring: a refill of the window never waits for LDS
    uint32_t* ring;              // this wave's ring (LDS), already offset by the lane
    __device__ __forceinline__ void request(uint32_t* wave_ring, int lane, uint32_t upto)      // dwords [loaded, upto), upto - next <= 16
    {
        vlc_request(base, loaded, upto, nd, wave_ring, lane);
        loaded = upto;
    }
    __device__ __forceinline__ uint32_t raw(uint32_t i) const { return ring[(i & 15u) * 64u]; }
    __device__ __forceinline__ uint32_t slot(uint32_t i) const { return __builtin_bswap32(raw(i)); }
    __device__ __forceinline__ void init(const uint32_t* b, uint32_t n_dwords, uint32_t bit_pos, uint32_t* wave_ring, int lane)
    {
        base = b; nd = n_dwords; pos = bit_pos;
        next = loaded = bit_pos >> 5;
        ring = wave_ring + lane;
        request(wave_ring, lane, next + kVlcRingDwords);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        safe = loaded;
        const uint32_t lead = bit_pos & 31u;
        const uint32_t d0 = slot(next), d1 = slot(next + 1u);
        next += 2u;
        ahead = raw(next);
        w = (((uint64_t)d0 << 32) | d1) << lead;
        avail = 64 - (int)lead;
    }
    // the next dword of the stream; its successor is fetched from the ring at once and is long there when it is wanted
    // (read where it is needed, the LDS latency sat on the path of every symbol: some lane of the 64 refills at every step)
    __device__ __forceinline__ uint32_t take()
    {
        const uint32_t d = __builtin_bswap32(ahead);      // swapped when it is used: the read below is not waited for here
        next++;
        ahead = raw(next);
        return d;
    }
    // Top-up point (wave-uniform): called in front of every syntax element; between two calls a lane takes at most one
    // dword.  When any lane is down to half a ring, or to its last landed dwords: wait for what is on its way
    // (requested half a ring ago), then request up to a full ring again.  (Requesting without the wait and waiting only
    // when a lane is about to read what has not been waited for -- so that the wait sees the stores of the last symbols
    // instead of the loads, gfx950 counts both on one counter -- measured no faster.)
    __device__ __forceinline__ void sync(uint32_t* wave_ring, int lane)
    {
        const bool low = loaded - next < 8u || safe - next < 3u;      // take() reads dword next + 1
        if (__builtin_amdgcn_ballot_w64(low) != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            safe = loaded;
            request(wave_ring, lane, next + kVlcRingDwords);
        }
    }
    // at least 32 valid bits afterwards (one symbol of the syntax takes at most 28)
    __device__ __forceinline__ void fill()
    {
        if (avail <= 32) {
            w |= (uint64_t)take() << (32 - avail);
            avail += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(w >> (64 - n)); }     // 1 <= n <= 32
    __device__ __forceinline__ void drop(int n) { w <<= n; avail -= n; pos += (uint32_t)n; }
    __device__ __forceinline__ uint32_t get(int n)
    {
        if (n == 0) return 0u;
        fill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

struct VlcLds {
    uint16_t fast12[4096];
    uint16_t long9[512];
    uint16_t motion_s[2048];
    uint16_t mba[2048];
    uint16_t cbp[512];
    uint16_t mbtype[4][64];
    uint16_t dc_lum[128], dc_chr[256];
    uint16_t zz_off[64];
};
static_assert(sizeof(VlcLds) == kVlcLdsWords * 4, "LDS copy and VlcTables disagree");

struct VlcCtx {                  // per lane: the state a slice carries from macroblock to macroblock
    int mb_addr, mb_row, mb_col, rc_addr;
    int fw_h, fw_v, fw_h_prev, fw_v_prev, bw_h, bw_v, bw_h_prev, bw_v_prev, prev_dir;
    int dc_y, dc_cr, dc_cb, dc_a, qs;
    int mb_intra;
    VLC_G uint32_t* hdr;         // next free block header of the scratch strip
    VLC_G uint32_t* ent;         // next free entry
    VLC_G uint32_t* hdr_end;
    VLC_G uint32_t* ent_end;
    VLC_G char* zbase;           // the picture's counters and maps
    int type, full_pel_fwd, fwd_rsize, full_pel_bwd, bwd_rsize;
    uint32_t* wave_ring;         // LDS ring of the wave (VlcWin::sync)
    int lane;
};
#define VLC_SYNC(r, c) (r).sync((c).wave_ring, (c).lane)

// decoders/jsv.js:831-893, as motion_component of leon_vlc.cpp
__device__ __forceinline__ int vlc_motion_component(VlcWin& r, const VlcLds& L, int prev, int rsize, int f, int& err)
{
    r.fill();
    const uint32_t e = L.motion_s[r.peek(11)];
    if (e == 0) { err = VLC_ERR_MOTION; return prev; }
    r.drop((int)(e >> 8));
    const int code = (int)(e & 0xffu) - 16;
    int d = code;
    if (code != 0 && f != 1) {
        const int res = (int)r.get(rsize);
        d = (((code < 0 ? -code : code) - 1) << rsize) + res + 1;
        if (code < 0) d = -d;
    }
    prev += d;
    if (prev > (f << 4) - 1) prev -= f << 5;
    else if (prev < -(f << 4)) prev += f << 5;
    return prev;
}

// decoders/jsv.js:1338-1525 (decodeBlockGL), as decode_block of leon_vlc.cpp; returns an error code or 0
// (All coded blocks of a macroblock in ONE loop -- an iteration = one symbol of whatever block the lane is in -- was
// tried twice, the second time with a block start of a dozen instructions, and is slower both times: 71 k against
// 93-95 k pictures/s end to end on one box.  The tight per-slot loop wins although it runs more iterations.)
// COMP (which DC predictor, which DC table): 0 luma blocks 0..3, 1 block 4, 2 block 5, 3 the A blocks 6..9 -- a template
// parameter: chosen at run time, the predictor would be read through a computed address and the whole context
// would live in scratch memory
template <int COMP>
__device__ __forceinline__ int vlc_block(VlcWin& r, const VlcLds& L, const VlcTables* __restrict__ T, const VlcGeom& G, VlcCtx& c, int block)
{
    uint32_t gid, bq;
    if (COMP == 0 || COMP == 3) {                            // luma, or the A component (blocks 6..9, placed like luma)
        const int lb = COMP == 0 ? block : block - 6;
        const int qb = c.mb_col * 2 + (lb & 1);
        gid = (uint32_t)((2 * c.mb_row + (lb >> 1)) * G.gy + (qb >> 3)) + (COMP == 0 ? 0u : (uint32_t)(G.n_y + 2 * G.n_c));
        bq = (uint32_t)(qb & 7);
    } else {
        gid = (uint32_t)(G.n_y + (COMP == 2 ? G.n_c : 0) + c.mb_row * G.gc + (c.mb_col >> 3));
        bq = (uint32_t)(c.mb_col & 7);
    }
    // an entry as k_vlc_parse leaves it: (block of the group * 64 + zig-zag index) << 16 | level; k_vlc_gather turns the
    // index into the tile offset of include/leon_vlc.h on its way (the table lookup was an LDS round trip per symbol here)
    const uint32_t bbase = (bq * 64u) << 16;
    if (c.hdr >= c.hdr_end || c.ent + 64 > c.ent_end) return VLC_ERR_SCRATCH;      // a block: one header, at most 64 entries
    VLC_G uint32_t* const rec = c.ent;
    int k = 0, n = 0;
    if (c.mb_intra) {
        VLC_SYNC(r, c);
        r.fill();
        const uint32_t e = COMP == 0 || COMP == 3 ? L.dc_lum[r.peek(7)] : L.dc_chr[r.peek(8)];
        if (e == 0) return VLC_ERR_DC;
        r.drop((int)(e >> 8));
        const int size = (int)(e & 0xffu);
        const int predictor = COMP == 0 ? c.dc_y : COMP == 3 ? c.dc_a : COMP == 1 ? c.dc_cr : c.dc_cb;
        int dc = predictor;
        if (size > 0) {
            VLC_SYNC(r, c);
            const int differential = (int)r.get(size);
            dc = (differential & (1 << (size - 1))) ? predictor + differential
                                                    : predictor + ((int)(0xffffffffu << size) | (differential + 1));
        }
        if (COMP == 0) c.dc_y = dc; else if (COMP == 3) c.dc_a = dc; else if (COMP == 1) c.dc_cr = dc; else c.dc_cb = dc;
        if ((int16_t)dc != 0) rec[k++] = bbase | (uint16_t)(int16_t)dc;
        n = 1;
    }
    // the first symbol of a non-intra block: '1s' is run 0, level +-1 (two bits), and there is no end-of-block code.
    // Every other first symbol starts with a 0 bit and reads like any later one: only this case is handled here, in
    // front of the loop (inside it, the test cost every iteration six instructions).
    if (n == 0) {
        VLC_SYNC(r, c);
        r.fill();
        const uint32_t p2 = r.peek(2);
        if (p2 & 2u) {
            r.drop(2);
            rec[k++] = bbase | (uint32_t)(uint16_t)(int16_t)((p2 & 1u) ? -1 : 1);      // zig-zag position 0
            n = 1;
        }
    }
    for (;;) {
        VLC_SYNC(r, c);
        r.fill();
        const uint32_t f = L.fast12[(uint32_t)(r.w >> 52)];
        const int flen = (int)(f & 0xfu);
        int run_len, level;
        if (flen) {
            r.drop(flen);
            if (f & 0x10u) break;                             // end of block
            run_len = (int)((f >> 5) & 31u);
            level = (int)(int16_t)(uint16_t)f >> 10;
        } else {
            // escapes ('0000 01', read arithmetically) and the codes of 12 .. 16 bits (seven zeros in front; a second
            // table in LDS).  (Through the 16-bit table in global memory, a wave waited for memory in one iteration
            // of six: some lane of the 64 is here.)
            const uint64_t w = r.w;
            int used;
            if ((w >> 58) == 1u) {                           // escape: 6-bit run, 8- or 16-bit level
                run_len = (int)((w >> 52) & 63);
                level = (int)((w >> 44) & 255);
                used = 20;
                if (level == 0) { level = (int)((w >> 36) & 255); used = 28; }
                else if (level == 128) { level = (int)((w >> 36) & 255) - 256; used = 28; }
                else if (level > 128) level -= 256;
            } else {
                if ((w >> 57) != 0u) return VLC_ERR_COEF;
                const uint32_t e = L.long9[(uint32_t)(w >> 48) & 511u];
                if (e == 0u) return VLC_ERR_COEF;
                const int len = (int)(e & 31u);
                run_len = (int)((e >> 5) & 31u);
                level = (int)(e >> 10);
                if ((w >> (63 - len)) & 1) level = -level;
                used = len + 1;
            }
            r.drop(used);
        }
        n += run_len;
        if (n > 63) return VLC_ERR_INDEX;
        if (level != 0) rec[k++] = bbase | ((uint32_t)n << 16) | (uint16_t)(int16_t)level;
        n++;
    }
    if (r.pos > (uint32_t)(r.nd << 5)) return VLC_ERR_END;     // ran off the data (zeros behind it: an invalid code ended the loop at the latest)
    if (k) {
        *c.hdr++ = (gid << 7) | (uint32_t)k;
        c.ent = rec + k;
        __hip_atomic_fetch_add(reinterpret_cast<VLC_G uint32_t*>(c.zbase + G.off_cnt) + gid, (uint32_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return 0;
}

__device__ __forceinline__ uint32_t vlc_mv_word(int h, int v) { return (uint32_t)(uint16_t)(int16_t)h | ((uint32_t)(uint16_t)(int16_t)v << 16); }

// macroblock_address_increment: 1..33, 34 = stuffing, 35 = escape, -1 = invalid code
__device__ __forceinline__ int vlc_mba(VlcWin& r, const VlcLds& L, VlcCtx& c)
{
    VLC_SYNC(r, c);
    r.fill();
    const uint32_t e = L.mba[r.peek(11)];
    if (e == 0) return -1;
    r.drop((int)(e >> 8));
    return (int)(e & 0xffu);
}

// decoders/jsv.js:725-828 (+ B pictures), as decode_macroblock of leon_vlc.cpp.  0 = macroblock read, 1 = stop
// silently (an address past the picture), > 1 an error
__device__ __forceinline__ int vlc_macroblock(VlcWin& r, const VlcLds& L, const VlcTables* __restrict__ T, const VlcGeom& G, VlcCtx& c,
                                              bool& slice_begin)
{
    const int type = c.type, mbsize = G.mbw * G.mbh;
    VLC_G uint8_t* const m_qscale = reinterpret_cast<VLC_G uint8_t*>(c.zbase + G.off_qscale);
    VLC_G uint8_t* const m_intra = reinterpret_cast<VLC_G uint8_t*>(c.zbase + G.off_intra);
    VLC_G uint8_t* const m_repadd = reinterpret_cast<VLC_G uint8_t*>(c.zbase + G.off_repadd);
    VLC_G uint8_t* const m_mb_dir = reinterpret_cast<VLC_G uint8_t*>(c.zbase + G.off_mb_dir);
    VLC_G uint32_t* const m_mv_fwd = reinterpret_cast<VLC_G uint32_t*>(c.zbase + G.off_mv_fwd);      // (h, v) int16 pairs as one word
    VLC_G uint32_t* const m_mv_bwd = reinterpret_cast<VLC_G uint32_t*>(c.zbase + G.off_mv_bwd);

    int increment = 0, t;
    t = vlc_mba(r, L, c);
    while (t == 34) t = vlc_mba(r, L, c);                    // stuffing
    while (t == 35) { increment += 33; t = vlc_mba(r, L, c); }   // escape
    if (t < 0) return VLC_ERR_MBA + 1;
    increment += t;
    if (slice_begin) {
        slice_begin = false;
        c.mb_addr += increment;
    } else {
        if (c.mb_addr + increment >= mbsize) return 1;
        if (increment > 1) {
            c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
            if (type == 2) { c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0; }
        }
        while (increment > 1) {                              // skipped macroblocks
            const int a = ++c.mb_addr;
            if (type != 1) m_mv_fwd[a] = vlc_mv_word(c.fw_h, c.fw_v);
            if (type == 3) {
                m_mv_bwd[a] = vlc_mv_word(c.bw_h, c.bw_v);
                m_mb_dir[a] = (uint8_t)c.prev_dir;
            }
            increment--;
        }
        c.mb_addr++;
    }
    const int mb = c.mb_addr;
    if (mb < 0 || mb >= mbsize) return VLC_ERR_ADDR + 1;
    c.mb_col += mb - c.rc_addr;
    c.rc_addr = mb;
    while (c.mb_col >= G.mbw) { c.mb_col -= G.mbw; c.mb_row++; }
    VLC_SYNC(r, c);
    r.fill();
    const uint32_t te = L.mbtype[type][r.peek(6)];
    if (te == 0) return VLC_ERR_TYPE + 1;
    r.drop((int)(te >> 8));
    const int mb_type = (int)(te & 0xffu);
    c.mb_intra = mb_type & 0x01;
    const int mot_fw = mb_type & 0x08, mot_bw = mb_type & 0x04;
    if (mb_type & 0x10) c.qs = (int)r.get(5);                 // type (<= 6 bits) and quantiser_scale: one dword at most
    m_qscale[mb] = (uint8_t)c.qs;
    m_intra[mb] = c.mb_intra ? 255 : 0;
    if (c.mb_intra) {
        c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0;
        c.bw_h = c.bw_h_prev = 0; c.bw_v = c.bw_v_prev = 0;
        c.prev_dir = 0;
        if (type != 1) m_repadd[mb] = 255;                   // jsv.js:1502-1505
    } else {
        c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
        int err = 0;
        if (mot_fw) {
            VLC_SYNC(r, c);
            c.fw_h_prev = vlc_motion_component(r, L, c.fw_h_prev, c.fwd_rsize, 1 << c.fwd_rsize, err);
            c.fw_h = c.full_pel_fwd ? c.fw_h_prev * 2 : c.fw_h_prev;
            VLC_SYNC(r, c);
            c.fw_v_prev = vlc_motion_component(r, L, c.fw_v_prev, c.fwd_rsize, 1 << c.fwd_rsize, err);
            c.fw_v = c.full_pel_fwd ? c.fw_v_prev * 2 : c.fw_v_prev;
        } else if (type == 2) {
            c.fw_h = c.fw_h_prev = 0;
            c.fw_v = c.fw_v_prev = 0;
        }
        if (mot_bw) {
            VLC_SYNC(r, c);
            c.bw_h_prev = vlc_motion_component(r, L, c.bw_h_prev, c.bwd_rsize, 1 << c.bwd_rsize, err);
            c.bw_h = c.full_pel_bwd ? c.bw_h_prev * 2 : c.bw_h_prev;
            VLC_SYNC(r, c);
            c.bw_v_prev = vlc_motion_component(r, L, c.bw_v_prev, c.bwd_rsize, 1 << c.bwd_rsize, err);
            c.bw_v = c.full_pel_bwd ? c.bw_v_prev * 2 : c.bw_v_prev;
        }
        if (err) return err + 1;
        if (type != 1) m_mv_fwd[mb] = vlc_mv_word(c.fw_h, c.fw_v);
        if (type == 3) {
            m_mv_bwd[mb] = vlc_mv_word(c.bw_h, c.bw_v);
            c.prev_dir = (mot_fw ? 1 : 0) | (mot_bw ? 2 : 0);
            m_mb_dir[mb] = (uint8_t)c.prev_dir;
        }
    }
    int cbp = 0;
    VLC_SYNC(r, c);
    if (mb_type & 0x02) {
        r.fill();
        const uint32_t ce = L.cbp[r.peek(9)];
        if (ce == 0) return VLC_ERR_CBP + 1;
        r.drop((int)(ce >> 8));
        cbp = (int)(ce & 0xffu);
    } else if (c.mb_intra) cbp = 0x3f;
    int apat = 0;
    if (G.alpha) apat = c.mb_intra ? 0xf : (int)r.get(4);     // pattern (<= 9 bits) and alpha_pattern: one dword at most
#pragma unroll 1
    for (int block = 0, mask = 0x20; block < 4; block++, mask >>= 1)
        if (cbp & mask) { const int e = vlc_block<0>(r, L, T, G, c, block); if (e) return e + 1; }
    if (cbp & 0x2) { const int e = vlc_block<1>(r, L, T, G, c, 4); if (e) return e + 1; }
    if (cbp & 0x1) { const int e = vlc_block<2>(r, L, T, G, c, 5); if (e) return e + 1; }
#pragma unroll 1
    for (int block = 6, mask = 0x8; block < 10; block++, mask >>= 1)
        if (apat & mask) { const int e = vlc_block<3>(r, L, T, G, c, block); if (e) return e + 1; }
    return 0;
}

// LDS (tables 18.4 KB + four rings of 4 KB per workgroup) allows four waves per SIMD: let the registers go that far too
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_vlc_parse(const VlcSlice* __restrict__ slices, uint32_t* __restrict__ slice_words, int n_slices,
                                                   const VlcPic* __restrict__ pics, uint32_t* __restrict__ errors, VlcGeom G,
                                                   const VlcTables* __restrict__ T)
{
    __shared__ VlcLds L;
    __shared__ __attribute__((aligned(16))) uint32_t rings[4 * kVlcRingDwords * 64];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(T);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&L);
        for (int i = threadIdx.x; i < kVlcLdsWords; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_slices) return;
    const VlcSlice S = slices[j];
    VlcCtx c{};
    c.lane = threadIdx.x & 63;
    c.wave_ring = rings + (threadIdx.x >> 6) * (kVlcRingDwords * 64);
    VlcWin r;
    r.init(S.bytes, S.n_dwords, S.bit_pos, c.wave_ring, c.lane);
    {
        const VlcPic* P = pics + S.pic;
        c.zbase = (VLC_G char*)P->zbase;
        c.type = P->type;
        c.full_pel_fwd = P->full_pel_fwd; c.fwd_rsize = P->fwd_rsize;
        c.full_pel_bwd = P->full_pel_bwd; c.bwd_rsize = P->bwd_rsize;
    }
    c.mb_addr = (S.code - 1) * G.mbw - 1;                     // decoders/jsv.js:683-706
    c.mb_row = S.code - 1;
    c.mb_col = -1;
    c.rc_addr = c.mb_addr;
    c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
    c.hdr = (VLC_G uint32_t*)S.scratch;
    c.hdr_end = c.ent = c.hdr + S.hdr_cap;
    c.ent_end = c.ent + S.ent_cap;
    c.qs = (int)r.get(5);
    for (;;) {                                                // extra_information_slice
        VLC_SYNC(r, c);
        if (!r.get(1) || r.pos >= (S.end_byte << 3)) break;
        r.get(8);
    }
    bool slice_begin = true;
    int err = 0;
    for (;;) {
        const int rc = vlc_macroblock(r, L, T, G, c, slice_begin);       // 1: the reference's silent return, the loop goes on
        if (rc > 1) { err = rc - 1; break; }
        // next_bits_are_start_code (decoders/jsv.js:1710-1760): byte aligned 00 00 01, or the end of the data
        VLC_SYNC(r, c);
        r.fill();
        const uint32_t i = (r.pos + 7u) >> 3, skip = (0u - r.pos) & 7u;
        if (i + 2u >= S.n_bytes) break;
        if ((uint32_t)((r.w << skip) >> 40) == 1u) break;
        if (i >= S.end_byte) { err = VLC_ERR_END; break; }    // behind the start code the host found: ran over it
    }
    ((VLC_G uint32_t*)slice_words)[j] = (uint32_t)(c.hdr - (VLC_G uint32_t*)S.scratch);       // coded blocks of the slice
    if (err) atomicCAS(errors + S.pic, 0u, (uint32_t)err | ((uint32_t)S.code << 8));     // rare: a generic atomic is fine here
}

// zeroes the regions the parser counts in and reports through (one per GOP of the window: the arenas are separate
// allocations), 16 bytes per thread and round -- one launch instead of a fill per GOP
struct VlcClear { uint4* ptr; uint64_t n16; };
__global__ __launch_bounds__(256) void k_vlc_clear(const VlcClear* __restrict__ regions)
{
    const VlcClear R = regions[blockIdx.y];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < R.n16; i += (uint64_t)gridDim.x * 256) R.ptr[i] = uint4{0u, 0u, 0u, 0u};
}

// exclusive scan of a picture's group counters; the counters go back to zero (k_vlc_gather's cursors)
__global__ __launch_bounds__(256) void k_vlc_offsets(const VlcPic* __restrict__ pics, VlcGeom G)
{
    __shared__ uint32_t part[256];
    const VlcPic P = pics[blockIdx.x];
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(P.zbase + G.off_cnt);
    const int ng = G.n_groups, per = (ng + 255) / 256;
    const int lo = min((int)threadIdx.x * per, ng), hi = min(lo + per, ng);
    uint32_t sum = 0;
    for (int g = lo; g < hi; g++) sum += cnt[g];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                       // inclusive scan of the 256 partial sums
        const uint32_t v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int g = lo; g < hi; g++) {
        const uint32_t n = cnt[g];
        P.grp_off[g] = run;
        cnt[g] = 0;
        run += n;
    }
    if (threadIdx.x == 255) P.grp_off[ng] = part[255];
}

__global__ __launch_bounds__(256) void k_vlc_gather(const VlcSlice* __restrict__ slices, const uint32_t* __restrict__ slice_blocks, int n_slices,
                                                    const VlcPic* __restrict__ pics, VlcGeom G, const VlcTables* __restrict__ T)
{
    __shared__ uint32_t s_from[4][65], s_at[4][64];                  // per wave: where the round's blocks start in the strip, where they go
    __shared__ uint16_t s_zz[64];                                    // zig-zag index -> byte offset of the coefficient in its block's part of the tile
    if (threadIdx.x < 64) s_zz[threadIdx.x] = T->zz_off[threadIdx.x];
    __syncthreads();
    const int wv = (int)(threadIdx.x >> 6);
    const int j = blockIdx.x * 4 + wv;                               // one wave per slice
    const int lane = threadIdx.x & 63;
    if (j >= n_slices) return;
    const VlcSlice S = slices[j];
    const VlcPic P = pics[S.pic];
    VLC_G uint32_t* const cursor = reinterpret_cast<VLC_G uint32_t*>((VLC_G char*)P.zbase + G.off_cnt);
    const uint32_t n_blocks = min(slice_blocks[j], S.hdr_cap);
    const VLC_G uint32_t* const hdr = (const VLC_G uint32_t*)S.scratch;
    const VLC_G uint32_t* const ent = hdr + S.hdr_cap;
    const VLC_G uint32_t* const grp_off = (const VLC_G uint32_t*)P.grp_off;
    VLC_G uint32_t* const entries = (VLC_G uint32_t*)P.entries;
    uint32_t carry = 0;                                              // entries of the blocks before this round
    for (uint32_t base = 0; base < n_blocks; base += 64) {
        const uint32_t b = base + (uint32_t)lane;
        const uint32_t h = b < n_blocks ? hdr[b] : 0u;
        const uint32_t gid = h >> 7, k = h & 127u;
        uint32_t incl = k;                                           // inclusive prefix sum of k over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t from = carry + incl - k;
        const uint32_t round_begin = carry;
        carry += (uint32_t)__shfl((int)incl, 63, 64);
        // Neighbouring lanes often hold the two blocks of one macroblock that lie in the same group (Y0 Y1, Y2 Y3):
        // such a pair takes ONE atomic add -- the launch is bound by their rate.  A lane follows its left neighbour
        // when both have entries for the same group and the neighbour does not follow somebody itself.
        const uint32_t gid_l = (uint32_t)__shfl_up((int)gid, 1, 64), k_l = (uint32_t)__shfl_up((int)k, 1, 64);
        const bool same_l = lane > 0 && k != 0u && k_l != 0u && gid_l == gid;
        const bool follows = same_l && !__shfl_up((int)same_l, 1, 64);
        const bool followed = __shfl_down((int)follows, 1, 64) != 0 && lane < 63;
        const uint32_t k_r = (uint32_t)__shfl_down((int)k, 1, 64);
        const bool ok = k != 0u && k <= 64u && from + k <= S.ent_cap;
        uint32_t at = 0u;
        if (ok && !follows) at = grp_off[gid] + __hip_atomic_fetch_add(cursor + gid, followed ? k + k_r : k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t at_l = (uint32_t)__shfl_up((int)at, 1, 64);
        if (follows) at = at_l + k_l;
        // The entries of the round's blocks lie one behind the other in the strip: the wave copies them 64 at a time,
        // contiguous reads, each lane finding the block its entry belongs to by a binary search over the blocks' start
        // offsets (in LDS).  A lane copying its own block entry by entry issued as many instructions as the longest
        // block of the wave is long, each with 64 addresses 4 bytes wide all over memory.
        s_from[wv][lane] = from;
        s_at[wv][lane] = ok && at + k <= P.entries_cap ? at : 0xffffffffu;
        if (lane == 63) s_from[wv][64] = carry;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t round_end = min(carry, S.ent_cap);
        for (uint32_t e = round_begin + (uint32_t)lane; e < round_end; e += 64u) {
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t step = 32; step != 0; step >>= 1)
                if (s_from[wv][lo + step] <= e) lo += step;
            const uint32_t dst = s_at[wv][lo];
            if (dst != 0xffffffffu) {
                // (block * 64 + zig-zag index) -> the tile offset block * 16 + zz_off[index]
                const uint32_t v = ent[e], hi = v >> 16;
                entries[dst + (e - s_from[wv][lo])] = ((((hi >> 6) & 7u) * 16u + s_zz[hi & 63u]) << 16) | (v & 0xffffu);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

}  // namespace leon
