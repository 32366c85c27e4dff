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
//   k_vlc_parse    one lane per slice, the SERIAL part only: macroblock headers and vectors (ONE 16-byte record per
//                  macroblock), DC values, and for every coded block WHERE its coefficient symbols begin and how
//                  many there are -- the symbols themselves are stepped over, several at a time (one table lookup on
//                  12 bits gives the bits, the number of symbols and the positions they advance, up to the end of
//                  block).  One 16-byte record per coded block into the picture's record array, the slice's records
//                  side by side from (first macroblock address x blocks per macroblock) on.  No atomics.
//   k_vlc_index    one workgroup per picture: counts the entries of every group from the block records (LDS atomics;
//                  a record learns its block's place inside the group), exclusive scan -> grp_off (what
//                  leon_sparse_picture wants), and turns the macroblock records into the maps (whole lines).
//   k_vlc_blocks   the PARALLEL part, one lane per coded block (a wave walks the records of one slice, 64 at a time):
//                  decodes the block's symbols from its bit position and writes the entries to grp_off[group] + place.
//                  Entries of a group are "in no particular order" (include/leon_vlc.h).
// Round 2 decoded the coefficients in the slice loop (one kernel + a gather): a launch lasted as long as its longest
// slice, symbol after symbol, 170 ns each.  A block's end can only be found by reading its symbols' LENGTHS in order --
// but nothing else of them is needed there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace leon {

// Pointers into device memory are told to be GLOBAL: through a generic pointer every store is a flat_store, which
// counts on the LDS counter too -- and the wait in front of the next table lookup (LDS) then waits for the entry
// stored a symbol earlier to reach memory.
#define VLC_G __attribute__((address_space(1)))
typedef uint32_t vlc_u4 __attribute__((ext_vector_type(4)));      // a 16-byte record (HIP's uint4 is a class: no address-space pointers to it)

#ifndef LEON_VLC_WAVES
#define LEON_VLC_WAVES 4
#endif
#ifndef LEON_VLC_LOCKSTEP
#define LEON_VLC_LOCKSTEP 0      // 1: the slice loop as a state machine whose rounds the lanes of a wave run in step (round 4: bit-exact, and slower -- see below); 0: round 3's loop
#endif
// (Round 4: s_setprio 3 for the waves of k_vlc_parse, of k_vlc_index / k_vlc_blocks, of all three -- beside reconstruction waves at 0:
// the same 187 k pictures/s end to end in every arrangement, and k_vlc_parse 4.80 ms per window against 4.87: the slice loop waits
// for its own dependent LDS lookups, not for issue slots.  Removed.)
#ifndef LEON_VLC_TWO_STEPS
#define LEON_VLC_TWO_STEPS 1
#endif
#ifndef LEON_VLC_MULTI_BITS
#define LEON_VLC_MULTI_BITS 12
#endif
static constexpr int kVlcMultiBits = LEON_VLC_MULTI_BITS;      // how many bits of the stream one step of k_vlc_parse's coefficient loop looks at
static constexpr int kVlcMulti = 1 << kVlcMultiBits;

// Device copy of the front end's tables (leon_vlc_get_gpu_tables), 16 bits per entry; k_vlc_parse's part first:
//   multi12: the coefficient symbols that lie COMPLETE in the next 12 bits, stepped over together: bits 0..3 the bits
//            they take (0 = the first one is a longer code or an escape), bits 4..6 how many symbols (the end-of-block
//            code not counted), bit 7 an end-of-block code is among them (the last one), bits 8..15 the sum of (run + 1)
//   long9:   the codes of 12 .. 16 bits (all of them start with seven zeros) by the nine bits behind those zeros:
//            bits 0..4 length without the sign bit, bits 5..9 run, bits 10..15 level (unsigned), 0 = invalid code
//   fast12:  one symbol from the next 12 bits (k_vlc_blocks): bits 0..3 length (0 = longer code or escape), bit 4 end of
//            block, bits 5..9 run, bits 10..15 level
//   the others: (length << 8) | value, 0 = invalid code
struct VlcTables {
    uint16_t multi12[kVlcMulti];
    uint16_t long9[512];
    uint16_t motion_s[2048];
    uint16_t mba[2048];
    uint16_t cbp[512];
    uint16_t mbtype[4][64];
    uint16_t dc_lum[128], dc_chr[256];
    uint16_t fast12[4096];       // k_vlc_blocks
    uint16_t zz_off[64];
};
static constexpr int kVlcLdsWords = (kVlcMulti + 512 + 2048 + 2048 + 512 + 256 + 128 + 256) / 2;
// every lane reads its slice through a ring of 16 dwords in LDS: dword i of lane L at ring[(i & 15) * 64 + L]
static constexpr int kVlcRingDwords = 16;
// a coded block as k_vlc_parse hands it to k_vlc_blocks: {bit position of its first coefficient symbol (behind the DC of an
// intra block), group | block of the group << 20 | intra << 23 | entries << 24, DC level | has one << 16, where its entries
// begin inside the group's list (filled in by k_vlc_index)} -- 16 bytes, one store
static constexpr int kVlcRecWords = 4;
// a macroblock as k_vlc_parse hands it to k_vlc_index: {quantiser scale | intra << 8 | RepAdd 255 << 9 | direction << 10,
// forward vector, backward vector, 0} -- ONE 16-byte store per macroblock into the picture's record array (index = macroblock
// address) instead of up to six scattered byte / word stores into six maps: the slice loop's stores leave the L2 as partial
// lines (a lane comes back to a line long after the reconstruction launches beside it have flushed it), round 3 counted 113 M
// write requests to memory per 1536-picture window, a tenth of them whole lines -- three times what the largest
// reconstruction launch writes -- and the kernels beside the parser ran at half speed
static constexpr int kVlcMbRecBytes = 16;
static constexpr int kVlcRingBytesPerWave = kVlcRingDwords * 64 * 4;

struct VlcGeom {
    int32_t mbw, mbh, gy, gc, n_y, n_c, n_groups, alpha;
    // byte offsets of a picture's maps from VlcPic::maps (the same for every picture: wave-uniform)
    uint32_t off_qscale, off_intra, off_repadd, off_mb_dir, off_mv_fwd, off_mv_bwd, pad0, pad1;
};

struct VlcPic {                  // one picture of the window
    int32_t type, full_pel_fwd, fwd_rsize, full_pel_bwd, bwd_rsize;
    uint32_t first_slice;        // its slices in the launch's slice array: [first_slice, first_slice + n_slices)
    char* zbase;                 // zero on entry: the macroblock records [mbw * mbh] of kVlcMbRecBytes
    char* maps;                  // the macroblock maps k_vlc_index writes, at VlcGeom's offsets
    uint32_t* grp_off;           // [n_groups + 1]
    uint32_t* entries;
    uint32_t entries_cap;
    uint32_t n_slices;
    uint32_t* recs;              // the block records of its slices: room for EVERY block of the picture (mbw * mbh * blocks per macroblock records)
};

struct VlcSlice {
    const uint32_t* bytes;       // the GOP's stream copy (4-byte aligned, zero padded)
    uint32_t n_dwords;           // readable dwords of it
    uint32_t n_bytes;            // its real length
    uint32_t bit_pos;            // first bit behind the slice start code
    uint32_t end_byte;           // first byte behind the slice
    int32_t code;                // slice_vertical_position
    uint32_t pic;
};

// Where a slice's block records lie (round 4).  Round 3 gave every slice a strip sized from its bytes (a coded block takes four
// bits at least: 32 bytes of records per byte of stream, 3.3 MB per 1080p picture, 60 % of the arenas).  But the slices of a
// picture do not overlap: slice s covers the macroblocks [first_s, last_s] with last_s < first_(s+1), and a macroblock has
// `bpm` blocks at most -- so a slice that writes its records side by side from record first_s * bpm on ends in front of record
// (last_s + 1) * bpm <= first_(s+1) * bpm, where the next slice begins: ONE array of mbs * bpm records per picture (0.78 MB at
// 1080p) holds them all, whatever the stream, and a lane knows where its part begins as soon as it has read its first
// macroblock address.  k_vlc_parse leaves {first record, end record, last macroblock address} per slice; k_vlc_index holds
// last_s < first_(s+1) against them -- slices that overlap (MPEG-1 forbids it; the host parser decodes them one after the
// other, the later one wins, as the reference does) would have written into each other's records, in an order nobody knows:
// the picture is refused (VLC_ERR_OVERLAP) instead.
struct VlcSliceOut { uint32_t first_rec, end_rec; int32_t last_mb; uint32_t pad; };
__host__ __device__ inline int vlc_blocks_per_mb(int alpha) { return alpha ? 10 : 6; }

enum { VLC_ERR_MBA = 1, VLC_ERR_ADDR, VLC_ERR_TYPE, VLC_ERR_MOTION, VLC_ERR_CBP, VLC_ERR_COEF, VLC_ERR_INDEX, VLC_ERR_END, VLC_ERR_DC,
       VLC_ERR_SCRATCH, VLC_ERR_OVERLAP };

// Requests dwords [loaded, upto) of a lane's stream into its ring slots: 16 conditional LDS-direct loads, one per
// slot (the LDS address of such a load is wave-uniform: slot i of every lane that wants it goes in one instruction).
// NOT inlined in the legacy loop: the top-up sits in front of every syntax element (28 places), and 28 copies of this made
// the kernel four times as long as the instruction cache.  The lockstep loop has ONE top-up per round: inlined there (a call
// in the middle of the loop costs the callee's register saves in scratch memory).
#if LEON_VLC_LOCKSTEP
__device__ __forceinline__ void vlc_request(
#else
__device__ __attribute__((noinline)) void vlc_request(
#endif
const uint32_t* base, uint32_t loaded, uint32_t upto, uint32_t nd, uint32_t* wave_ring, int lane)
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
    // MARGIN: how many dwords [next, next + MARGIN) must have landed afterwards -- the legacy loop takes at most one dword
    // between two calls (3: take() reads dword next + 1), the lockstep loop up to five in one round (7)
    template <int MARGIN = 3>
    __device__ __forceinline__ void sync(uint32_t* wave_ring, int lane, bool live = true)
    {
        const bool low = live && (loaded - next < (uint32_t)(MARGIN + 5) || safe - next < (uint32_t)MARGIN);
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
    uint16_t multi12[kVlcMulti];
    uint16_t long9[512];
    uint16_t motion_s[2048];
    uint16_t mba[2048];
    uint16_t cbp[512];
    uint16_t mbtype[4][64];
    uint16_t dc_lum[128], dc_chr[256];
};
static_assert(sizeof(VlcLds) == kVlcLdsWords * 4, "LDS copy and VlcTables disagree");

struct VlcCtx {                  // per lane: the state a slice carries from macroblock to macroblock
    int mb_addr, mb_row, mb_col, rc_addr;
    int fw_h, fw_v, fw_h_prev, fw_v_prev, bw_h, bw_v, bw_h_prev, bw_v_prev, prev_dir;
    int dc_y, dc_cr, dc_cb, dc_a, qs;
    int mb_intra;
    VLC_G uint32_t* hdr;         // next free block record of the slice (nullptr until its first macroblock address is known)
    VLC_G uint32_t* hdr_end;     // end of the picture's record array
    VLC_G char* zbase;           // the picture's macroblock records
    int type, full_pel_fwd, fwd_rsize, full_pel_bwd, bwd_rsize;
    uint32_t* wave_ring;         // LDS ring of the wave (VlcWin::sync)
    int lane;
    uint32_t n_bytes, end_byte;  // lockstep loop: the slice's bounds for the end-of-slice test
    const VlcPic* pics;          // (wave-uniform) for the one read of the picture's record array, at the slice's first macroblock
    VLC_G uint32_t* outs;        // (wave-uniform) the launch's VlcSliceOut array
    uint32_t pic;
};

// the slice's first macroblock address is known: its block records begin at record mb * (blocks per macroblock) of the
// picture's array (see VlcSliceOut)
__device__ __forceinline__ void vlc_first_macroblock(VlcCtx& c, const VlcGeom& G, int mb)
{
    const uint32_t bpm = (uint32_t)vlc_blocks_per_mb(G.alpha);
    VLC_G uint32_t* const recs = (VLC_G uint32_t*)c.pics[c.pic].recs;
    c.hdr = recs + (uint32_t)mb * bpm * kVlcRecWords;
    c.hdr_end = recs + (uint32_t)(G.mbw * G.mbh) * bpm * kVlcRecWords;
    c.outs[(blockIdx.x * 256u + threadIdx.x) * 4u] = (uint32_t)mb * bpm;
}
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

// decoders/jsv.js:1338-1525 (decodeBlockGL), the serial half: the DC value, then the block's coefficient symbols are
// STEPPED OVER -- their lengths decide where the block ends, nothing else of them is needed here (k_vlc_blocks reads them
// again, a lane per block).  One lookup on the next 12 bits covers every symbol that lies complete in them: the bits they
// take, how many they are, the positions they advance, and whether the last one is the end-of-block code.  Same error
// conditions as decode_block of leon_vlc.cpp: an invalid code, a position past 63, the end of the data.
// COMP (which DC predictor, which DC table): 0 luma blocks 0..3, 1 block 4, 2 block 5, 3 the A blocks 6..9 -- a template
// parameter: chosen at run time, the predictor would be read through a computed address and the whole context
// would live in scratch memory
template <int COMP>
__device__ __forceinline__ int vlc_block(VlcWin& r, const VlcLds& L, const VlcGeom& G, VlcCtx& c, int block)
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
    if (c.hdr + kVlcRecWords > c.hdr_end) return VLC_ERR_SCRATCH;
    int k = 0, n = 0;
    uint32_t dcw = 0u;
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
        if ((int16_t)dc != 0) { dcw = 0x10000u | (uint32_t)(uint16_t)(int16_t)dc; k = 1; }
        n = 1;
    }
    const uint32_t start = r.pos;                            // the block's coefficient symbols begin here
    // the first symbol of a non-intra block: '1s' is run 0, level +-1 (two bits), and there is no end-of-block code.
    // Every other first symbol starts with a 0 bit and reads like any later one.
    if (n == 0) {
        VLC_SYNC(r, c);
        r.fill();
        if (r.peek(2) & 2u) { r.drop(2); k = 1; n = 1; }
    }
    for (;;) {
        VLC_SYNC(r, c);
        r.fill();
        uint32_t m = L.multi12[(uint32_t)(r.w >> (64 - kVlcMultiBits))];
        int used = (int)(m & 15u);
        if (used) {
            r.drop(used);
            k += (int)((m >> 4) & 7u);
            n += (int)(m >> 8);                              // every symbol: its run, and the position it fills
            if (n > 64) return VLC_ERR_INDEX;                // = the last symbol's position past 63 (the earlier ones lie below it)
            if (m & 0x80u) break;                            // end of block
#if LEON_VLC_TWO_STEPS
            // a second step on the bits that are left: fill() left 33 or more, the first step took 12 at most, the
            // lookup wants 12.  What it cannot answer (an escape, a long code) waits for the top of the loop.
            m = L.multi12[(uint32_t)(r.w >> (64 - kVlcMultiBits))];
            used = (int)(m & 15u);
            if (used) {
                r.drop(used);
                k += (int)((m >> 4) & 7u);
                n += (int)(m >> 8);
                if (n > 64) return VLC_ERR_INDEX;
                if (m & 0x80u) break;
            }
#endif
        } else {
            // one symbol: an escape ('0000 01', 20 or 28 bits) or a code of 12 .. 16 bits (seven zeros in front; a
            // second table in LDS)
            const uint64_t w = r.w;
            int used1, run_len;
            if ((w >> 58) == 1u) {
                run_len = (int)((w >> 52) & 63);
                const uint32_t l8 = (uint32_t)(w >> 44) & 255u;
                used1 = (l8 == 0u || l8 == 128u) ? 28 : 20;
            } else {
                if ((w >> 57) != 0u) return VLC_ERR_COEF;
                const uint32_t e = L.long9[(uint32_t)(w >> 48) & 511u];
                if (e == 0u) return VLC_ERR_COEF;
                run_len = (int)((e >> 5) & 31u);
                used1 = (int)(e & 31u) + 1;
            }
            r.drop(used1);
            n += run_len;
            if (n > 63) return VLC_ERR_INDEX;
            n++;
            k++;
        }
    }
    if (r.pos > (uint32_t)(r.nd << 5)) return VLC_ERR_END;     // ran off the data (zeros behind it: an invalid code ended the loop at the latest)
    if (k) {
        *reinterpret_cast<VLC_G vlc_u4*>(c.hdr) = vlc_u4{start, gid | (bq << 20) | (c.mb_intra ? 1u << 23 : 0u) | ((uint32_t)k << 24), dcw, 0u};
        c.hdr += kVlcRecWords;
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
__device__ __forceinline__ int vlc_macroblock(VlcWin& r, const VlcLds& L, const VlcGeom& G, VlcCtx& c,
                                              bool& slice_begin)
{
    const int type = c.type, mbsize = G.mbw * G.mbh;
    VLC_G vlc_u4* const mbrec = reinterpret_cast<VLC_G vlc_u4*>(c.zbase);        // kVlcMbRecBytes per macroblock

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
            const int a = ++c.mb_addr;                          // (c.bw_*, c.prev_dir stay 0 outside B pictures)
            mbrec[a] = vlc_u4{(uint32_t)c.prev_dir << 10, vlc_mv_word(c.fw_h, c.fw_v), vlc_mv_word(c.bw_h, c.bw_v), 0u};
            increment--;
        }
        c.mb_addr++;
    }
    const int mb = c.mb_addr;
    if (mb < 0 || mb >= mbsize) return VLC_ERR_ADDR + 1;
    if (c.hdr == nullptr) vlc_first_macroblock(c, G, mb);
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
    if (c.mb_intra) {
        c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0;
        c.bw_h = c.bw_h_prev = 0; c.bw_v = c.bw_v_prev = 0;
        c.prev_dir = 0;
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
        if (type == 3) c.prev_dir = (mot_fw ? 1 : 0) | (mot_bw ? 2 : 0);
    }
    // quantiser scale | intra | RepAdd 255 of an intra macroblock outside I pictures (jsv.js:1502-1505) | direction, vectors
    mbrec[mb] = vlc_u4{(uint32_t)(c.qs & 0xff) | (c.mb_intra ? (type != 1 ? 0x300u : 0x100u) : (uint32_t)c.prev_dir << 10),
                      vlc_mv_word(c.fw_h, c.fw_v), vlc_mv_word(c.bw_h, c.bw_v), 0u};
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
        if (cbp & mask) { const int e = vlc_block<0>(r, L, G, c, block); if (e) return e + 1; }
    if (cbp & 0x2) { const int e = vlc_block<1>(r, L, G, c, 4); if (e) return e + 1; }
    if (cbp & 0x1) { const int e = vlc_block<2>(r, L, G, c, 5); if (e) return e + 1; }
#pragma unroll 1
    for (int block = 6, mask = 0x8; block < 10; block++, mask >>= 1)
        if (apat & mask) { const int e = vlc_block<3>(r, L, G, c, block); if (e) return e + 1; }
    return 0;
}

#ifndef LEON_VLC_HDR_QUORUM
#define LEON_VLC_HDR_QUORUM 12
#endif
// ---- the slice loop in LOCKSTEP (round 4) -------------------------------------------------------------------------------------
// Round 3's loop is the host parser's control flow, one lane per slice: decode_macroblock calls a block routine per coded
// block, each with a coefficient loop of its own -- six inlined copies of the same loop at six places of the program, around
// them the header code.  A wave executes the union of its lanes' paths: a lane stepping over the symbols of its Cb block and a
// lane stepping over those of a luma block run the SAME instructions at DIFFERENT addresses, one after the other; a lane that
// reads a macroblock header stalls 63 lanes that are in coefficients.  Measured: 10.8 of 64 lanes active per vector
// instruction (profiles/r03d_gpu_parser_sq_counters.txt), 508 k vector instructions per wave and window.
// Here a lane is a little state machine -- AT A MACROBLOCK BOUNDARY (end-of-slice test, address increment, header) ->
// AT A BLOCK (which one, DC value / first symbol) -> IN COEFFICIENTS (symbols stepped over, two table lookups a round) --
// and the wave runs rounds: every round has ONE place for each kind of work, executed by all the lanes that are in that
// state.  The header section is the long one and the rarest state; it is entered only when a quorum of lanes waits at a
// macroblock boundary (LEON_VLC_HDR_QUORUM), or when nobody has coefficients left to step over -- the macroblock headers
// are read in passes of their own, and the coefficient stepping in between runs lanes of like work.
// Same bits, same tables, same decisions and error conditions as the legacy loop (kept below, LEON_VLC_LOCKSTEP=0, for A/B
// runs) and as leon_vlc.cpp; tests/test_gpu_parser_gpu.py holds them against each other tensor by tensor.
enum { VLC_PH_MB = 0, VLC_PH_BLOCK = 1, VLC_PH_COEF = 2, VLC_PH_DONE = 3 };

struct VlcLane {                 // what a lane carries from round to round beside VlcCtx
    int phase;
    uint32_t todo;               // blocks of the macroblock still to read: bit (9 - block)
    int incr;                    // macroblock_address_increment collected so far (escapes), -1: none pending
    int mba_state;               // 0: stuffing may still come, 1: escapes only (the reference's two loops, jsv.js:737-748)
    uint32_t start, dcw, gidbq;  // the block being read: first coefficient bit, DC word, group | block of the group << 20
    int k, n;                    // entries so far, next coefficient position
    int err;
};

__device__ __forceinline__ void vlc_after_macroblock(VlcWin& r, const VlcCtx& S, VlcLane& l)
{
    // next_bits_are_start_code (decoders/jsv.js:1710-1760): byte aligned 00 00 01, or the end of the data
    r.fill();
    const uint32_t i = (r.pos + 7u) >> 3, skip = (0u - r.pos) & 7u;
    if (i + 2u >= S.n_bytes) { l.phase = VLC_PH_DONE; return; }
    if ((uint32_t)((r.w << skip) >> 40) == 1u) { l.phase = VLC_PH_DONE; return; }
    if (i >= S.end_byte) { l.err = VLC_ERR_END; l.phase = VLC_PH_DONE; }      // behind the start code the host found: ran over it
}

// one round of the macroblock-boundary state: ONE address increment code; when it is the last one (no stuffing, no escape),
// the rest of the header.  At most 11 + 11 + 4 * 17 + 13 = 103 bits.
__device__ __forceinline__ void vlc_round_header(VlcWin& r, const VlcLds& L, const VlcGeom& G, VlcCtx& c, VlcLane& l, bool& slice_begin)
{
    const int type = c.type, mbsize = G.mbw * G.mbh;
    VLC_G vlc_u4* const mbrec = reinterpret_cast<VLC_G vlc_u4*>(c.zbase);
    r.fill();
    const uint32_t e = L.mba[r.peek(11)];
    if (e == 0) { l.err = VLC_ERR_MBA; l.phase = VLC_PH_DONE; return; }
    r.drop((int)(e >> 8));
    const int t = (int)(e & 0xffu);
    if (t == 34 && l.mba_state == 0) return;                       // stuffing: the next code next round
    l.mba_state = 1;
    if (t == 35) { l.incr += 33; return; }                         // escape
    int increment = l.incr + t;
    l.incr = 0;
    l.mba_state = 0;
    if (slice_begin) {
        slice_begin = false;
        c.mb_addr += increment;
    } else {
        if (c.mb_addr + increment >= mbsize) { vlc_after_macroblock(r, c, l); return; }     // the reference's silent return: the loop goes on
        if (increment > 1) {
            c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
            if (type == 2) { c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0; }
        }
        while (increment > 1) {                                    // skipped macroblocks
            const int a = ++c.mb_addr;
            mbrec[a] = vlc_u4{(uint32_t)c.prev_dir << 10, vlc_mv_word(c.fw_h, c.fw_v), vlc_mv_word(c.bw_h, c.bw_v), 0u};
            increment--;
        }
        c.mb_addr++;
    }
    const int mb = c.mb_addr;
    if (mb < 0 || mb >= mbsize) { l.err = VLC_ERR_ADDR; l.phase = VLC_PH_DONE; return; }
    if (c.hdr == nullptr) vlc_first_macroblock(c, G, mb);
    c.mb_col += mb - c.rc_addr;
    c.rc_addr = mb;
    while (c.mb_col >= G.mbw) { c.mb_col -= G.mbw; c.mb_row++; }
    r.fill();
    const uint32_t te = L.mbtype[type][r.peek(6)];
    if (te == 0) { l.err = VLC_ERR_TYPE; l.phase = VLC_PH_DONE; return; }
    r.drop((int)(te >> 8));
    const int mb_type = (int)(te & 0xffu);
    c.mb_intra = mb_type & 0x01;
    const int mot_fw = mb_type & 0x08, mot_bw = mb_type & 0x04;
    if (mb_type & 0x10) c.qs = (int)r.get(5);
    if (c.mb_intra) {
        c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0;
        c.bw_h = c.bw_h_prev = 0; c.bw_v = c.bw_v_prev = 0;
        c.prev_dir = 0;
    } else {
        c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
        int err = 0;
        if (mot_fw) {
            c.fw_h_prev = vlc_motion_component(r, L, c.fw_h_prev, c.fwd_rsize, 1 << c.fwd_rsize, err);
            c.fw_h = c.full_pel_fwd ? c.fw_h_prev * 2 : c.fw_h_prev;
            c.fw_v_prev = vlc_motion_component(r, L, c.fw_v_prev, c.fwd_rsize, 1 << c.fwd_rsize, err);
            c.fw_v = c.full_pel_fwd ? c.fw_v_prev * 2 : c.fw_v_prev;
        } else if (type == 2) {
            c.fw_h = c.fw_h_prev = 0;
            c.fw_v = c.fw_v_prev = 0;
        }
        if (mot_bw) {
            c.bw_h_prev = vlc_motion_component(r, L, c.bw_h_prev, c.bwd_rsize, 1 << c.bwd_rsize, err);
            c.bw_h = c.full_pel_bwd ? c.bw_h_prev * 2 : c.bw_h_prev;
            c.bw_v_prev = vlc_motion_component(r, L, c.bw_v_prev, c.bwd_rsize, 1 << c.bwd_rsize, err);
            c.bw_v = c.full_pel_bwd ? c.bw_v_prev * 2 : c.bw_v_prev;
        }
        if (err) { l.err = err; l.phase = VLC_PH_DONE; return; }
        if (type == 3) c.prev_dir = (mot_fw ? 1 : 0) | (mot_bw ? 2 : 0);
    }
    mbrec[mb] = vlc_u4{(uint32_t)(c.qs & 0xff) | (c.mb_intra ? (type != 1 ? 0x300u : 0x100u) : (uint32_t)c.prev_dir << 10),
                      vlc_mv_word(c.fw_h, c.fw_v), vlc_mv_word(c.bw_h, c.bw_v), 0u};
    int cbp = 0;
    if (mb_type & 0x02) {
        r.fill();
        const uint32_t ce = L.cbp[r.peek(9)];
        if (ce == 0) { l.err = VLC_ERR_CBP; l.phase = VLC_PH_DONE; return; }
        r.drop((int)(ce >> 8));
        cbp = (int)(ce & 0xffu);
    } else if (c.mb_intra) cbp = 0x3f;
    int apat = 0;
    if (G.alpha) apat = c.mb_intra ? 0xf : (int)r.get(4);
    l.todo = ((uint32_t)cbp << 4) | (uint32_t)apat;                // bit 9 = block 0 ... bit 0 = block 9
    if (l.todo) l.phase = VLC_PH_BLOCK;
    else vlc_after_macroblock(r, c, l);
}

// the block state: which block, where its entries go, its DC value (intra) or its '1s' first symbol.  At most 16 bits.
__device__ __forceinline__ void vlc_round_block(VlcWin& r, const VlcLds& L, const VlcGeom& G, VlcCtx& c, VlcLane& l)
{
    const int block = __builtin_clz(l.todo) - 22;                  // the highest bit of the 10 that is set
    l.todo &= ~(0x200u >> block);
    uint32_t gid, bq;
    if (block < 4 || block >= 6) {                                 // luma, or the A component (blocks 6..9, placed like luma)
        const int lb = block < 4 ? block : block - 6;
        const int qb = c.mb_col * 2 + (lb & 1);
        gid = (uint32_t)((2 * c.mb_row + (lb >> 1)) * G.gy + (qb >> 3)) + (block < 4 ? 0u : (uint32_t)(G.n_y + 2 * G.n_c));
        bq = (uint32_t)(qb & 7);
    } else {
        gid = (uint32_t)(G.n_y + (block == 5 ? G.n_c : 0) + c.mb_row * G.gc + (c.mb_col >> 3));
        bq = (uint32_t)(c.mb_col & 7);
    }
    l.gidbq = gid | (bq << 20);
    if (c.hdr + kVlcRecWords > c.hdr_end) { l.err = VLC_ERR_SCRATCH; l.phase = VLC_PH_DONE; return; }
    l.k = 0; l.n = 0; l.dcw = 0u;
    if (c.mb_intra) {
        const bool lum = block < 4 || block >= 6;
        r.fill();
        const uint32_t e = lum ? L.dc_lum[r.peek(7)] : L.dc_chr[r.peek(8)];
        if (e == 0) { l.err = VLC_ERR_DC; l.phase = VLC_PH_DONE; return; }
        r.drop((int)(e >> 8));
        const int size = (int)(e & 0xffu);
        // (selects of VALUES, reads and writes alike: a predictor picked through a selected ADDRESS -- what the compiler makes of
        // conditional stores to different fields -- puts the whole context into scratch memory)
        const int py = c.dc_y, pcr = c.dc_cr, pcb = c.dc_cb, pa = c.dc_a;
        const int predictor = block < 4 ? py : block == 4 ? pcr : block == 5 ? pcb : pa;
        int dc = predictor;
        if (size > 0) {
            const int differential = (int)r.get(size);
            dc = (differential & (1 << (size - 1))) ? predictor + differential
                                                    : predictor + ((int)(0xffffffffu << size) | (differential + 1));
        }
        c.dc_y = block < 4 ? dc : py;
        c.dc_cr = block == 4 ? dc : pcr;
        c.dc_cb = block == 5 ? dc : pcb;
        c.dc_a = block >= 6 ? dc : pa;
        if ((int16_t)dc != 0) { l.dcw = 0x10000u | (uint32_t)(uint16_t)(int16_t)dc; l.k = 1; }
        l.n = 1;
        l.start = r.pos;
    } else {
        l.start = r.pos;
        r.fill();
        if (r.peek(2) & 2u) { r.drop(2); l.k = 1; l.n = 1; }       // '1s': run 0, level +-1; every other first symbol reads like a later one
    }
    l.phase = VLC_PH_COEF;
}

// the coefficient state: the block's symbols are stepped over (lengths only), two table lookups a round.  At most 28 bits.
__device__ __forceinline__ void vlc_round_coef(VlcWin& r, const VlcLds& L, VlcCtx& c, VlcLane& l)
{
    r.fill();
    bool eob = false;
    uint32_t m = L.multi12[(uint32_t)(r.w >> (64 - kVlcMultiBits))];
    int used = (int)(m & 15u);
    if (used) {
        r.drop(used);
        l.k += (int)((m >> 4) & 7u);
        l.n += (int)(m >> 8);
        if (l.n > 64) { l.err = VLC_ERR_INDEX; l.phase = VLC_PH_DONE; return; }
        eob = (m & 0x80u) != 0u;
        if (!eob) {      // a second step on the bits that are left (fill() left 33 or more, the first took 12 at most)
            m = L.multi12[(uint32_t)(r.w >> (64 - kVlcMultiBits))];
            used = (int)(m & 15u);
            if (used) {
                r.drop(used);
                l.k += (int)((m >> 4) & 7u);
                l.n += (int)(m >> 8);
                if (l.n > 64) { l.err = VLC_ERR_INDEX; l.phase = VLC_PH_DONE; return; }
                eob = (m & 0x80u) != 0u;
            }
        }
    } else {
        // one symbol: an escape ('0000 01', 20 or 28 bits) or a code of 12 .. 16 bits (seven zeros in front)
        const uint64_t w = r.w;
        int used1, run_len;
        if ((w >> 58) == 1u) {
            run_len = (int)((w >> 52) & 63);
            const uint32_t l8 = (uint32_t)(w >> 44) & 255u;
            used1 = (l8 == 0u || l8 == 128u) ? 28 : 20;
        } else {
            if ((w >> 57) != 0u) { l.err = VLC_ERR_COEF; l.phase = VLC_PH_DONE; return; }
            const uint32_t e = L.long9[(uint32_t)(w >> 48) & 511u];
            if (e == 0u) { l.err = VLC_ERR_COEF; l.phase = VLC_PH_DONE; return; }
            run_len = (int)((e >> 5) & 31u);
            used1 = (int)(e & 31u) + 1;
        }
        r.drop(used1);
        l.n += run_len;
        if (l.n > 63) { l.err = VLC_ERR_INDEX; l.phase = VLC_PH_DONE; return; }
        l.n++;
        l.k++;
    }
    if (!eob) return;
    if (r.pos > (uint32_t)(r.nd << 5)) { l.err = VLC_ERR_END; l.phase = VLC_PH_DONE; return; }     // ran off the data
    if (l.k) {
        *reinterpret_cast<VLC_G vlc_u4*>(c.hdr) = vlc_u4{l.start, l.gidbq | (c.mb_intra ? 1u << 23 : 0u) | ((uint32_t)l.k << 24), l.dcw, 0u};
        c.hdr += kVlcRecWords;
    }
    if (l.todo) l.phase = VLC_PH_BLOCK;
    else { l.phase = VLC_PH_MB; vlc_after_macroblock(r, c, l); }
}

// LDS (tables 18.4 KB + four rings of 4 KB per workgroup) allows four waves per SIMD: let the registers go that far too
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LEON_VLC_WAVES, LEON_VLC_WAVES))) void k_vlc_parse(const VlcSlice* __restrict__ slices, VlcSliceOut* __restrict__ slice_out, int n_slices,
                                                   const VlcPic* __restrict__ pics, uint32_t* __restrict__ errors, VlcGeom G,
                                                   const VlcTables* __restrict__ T)
{
    __shared__ VlcLds L;
    extern __shared__ __attribute__((aligned(16))) uint32_t rings[];          // 4 * kVlcRingDwords * 64 dwords, given at the launch
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
    c.hdr = c.hdr_end = nullptr;                              // until the first macroblock address is known (vlc_first_macroblock)
    c.pics = pics;
    c.outs = (VLC_G uint32_t*)slice_out;
    c.pic = S.pic;
    c.qs = (int)r.get(5);
    for (;;) {                                                // extra_information_slice
        VLC_SYNC(r, c);
        if (!r.get(1) || r.pos >= (S.end_byte << 3)) break;
        r.get(8);
    }
    bool slice_begin = true;
    int err = 0;
#if LEON_VLC_LOCKSTEP
    c.n_bytes = S.n_bytes;
    c.end_byte = S.end_byte;
    VlcLane l{};
    l.phase = VLC_PH_MB;
    // A wave whose lanes have all finished leaves; a lane without a slice has returned above (the wave's rounds are
    // decided by ballots over the lanes that are still here).
    for (;;) {
        const uint64_t live = __builtin_amdgcn_ballot_w64(l.phase != VLC_PH_DONE);
        if (live == 0) break;
        // up to 103 + 16 + 28 bits in a round: five dwords beside the one in hand
        r.sync<7>(c.wave_ring, c.lane, l.phase != VLC_PH_DONE);
        const uint64_t at_mb = __builtin_amdgcn_ballot_w64(l.phase == VLC_PH_MB);
        const uint64_t busy = __builtin_amdgcn_ballot_w64(l.phase == VLC_PH_BLOCK || l.phase == VLC_PH_COEF);
        // headers in passes of their own: when a quorum waits, or when nobody is left in coefficients
        if (at_mb != 0 && (busy == 0 || __builtin_popcountll(at_mb) >= LEON_VLC_HDR_QUORUM)) {
            if (l.phase == VLC_PH_MB) vlc_round_header(r, L, G, c, l, slice_begin);
        }
        if (__builtin_amdgcn_ballot_w64(l.phase == VLC_PH_BLOCK) != 0) {
            if (l.phase == VLC_PH_BLOCK) vlc_round_block(r, L, G, c, l);
        }
        if (l.phase == VLC_PH_COEF) vlc_round_coef(r, L, c, l);
    }
    err = l.err;
#else
    for (;;) {
        const int rc = vlc_macroblock(r, L, G, c, slice_begin);       // 1: the reference's silent return, the loop goes on
        if (rc > 1) { err = rc - 1; break; }
        // next_bits_are_start_code (decoders/jsv.js:1710-1760): byte aligned 00 00 01, or the end of the data
        VLC_SYNC(r, c);
        r.fill();
        const uint32_t i = (r.pos + 7u) >> 3, skip = (0u - r.pos) & 7u;
        if (i + 2u >= S.n_bytes) break;
        if ((uint32_t)((r.w << skip) >> 40) == 1u) break;
        if (i >= S.end_byte) { err = VLC_ERR_END; break; }    // behind the start code the host found: ran over it
    }
#endif
    {   // where the slice's records end, and its last macroblock address (VlcSliceOut; .first_rec was written at its first macroblock)
        VLC_G uint32_t* const o = (VLC_G uint32_t*)slice_out + (uint32_t)j * 4u;
        if (c.hdr != nullptr) {
            o[1] = (uint32_t)(G.mbw * G.mbh * vlc_blocks_per_mb(G.alpha)) - (uint32_t)(c.hdr_end - c.hdr) / kVlcRecWords;
            o[2] = (uint32_t)c.mb_addr;
        } else {
            o[0] = 0u; o[1] = 0u; o[2] = 0xffffffffu;          // never got to a macroblock (an error word says why)
        }
    }
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

// Between the two passes, a workgroup per picture:
//  * the entries of every group are counted from the block records of the picture's slices -- in LDS, so that neither pass
//    needs a global atomic (agent-scope atomics are write-through on a chip with one L2 per XCD: round 3 counted one
//    uncached 32-byte write per coded block, 32 M per window, in k_vlc_parse and again in k_vlc_blocks) -- and every
//    record learns where its block's entries begin inside the group's list (the counter's value before it was added);
//  * exclusive scan of the counters -> grp_off;
//  * the macroblock records become the maps the reconstruction reads (qscale, intra, RepAdd, direction, vectors), lane
//    after lane along the macroblock address: whole lines.
// Dynamic LDS: n_groups counters + a word per wave.
static constexpr int kVlcIndexThreads = 256;      // small workgroups: beside the reconstruction launches a 1024-thread one waited for 16 free wave slots on ONE CU
__global__ __launch_bounds__(kVlcIndexThreads) void k_vlc_index(const VlcSlice* __restrict__ slices, const VlcSliceOut* __restrict__ slice_out,
                                                                const VlcPic* __restrict__ pics, uint32_t* __restrict__ errors, VlcGeom G)
{
    extern __shared__ uint32_t s_index[];
    const int ng = G.n_groups, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* const cnt = s_index;
    uint32_t* const part = s_index + ng;
    const VlcPic P = pics[blockIdx.x];
    const uint32_t bpm = (uint32_t)vlc_blocks_per_mb(G.alpha), rec_cap = (uint32_t)(G.mbw * G.mbh) * bpm;
    for (int g = tid; g < ng; g += kVlcIndexThreads) cnt[g] = 0u;
    __syncthreads();
    // a wave per slice, a lane per record, four records of a lane in flight
    for (uint32_t sidx = (uint32_t)wv; sidx < P.n_slices; sidx += kVlcIndexThreads / 64) {
        const VlcSliceOut O = slice_out[P.first_slice + sidx];
        const uint32_t n_blocks = O.first_rec <= O.end_rec && O.end_rec <= rec_cap ? O.end_rec - O.first_rec : 0u;
        uint32_t* const rec = P.recs + (size_t)O.first_rec * kVlcRecWords;
        // the slices of a picture do not overlap (VlcSliceOut): the next one begins behind this one's last macroblock
        if (lane == 0 && sidx + 1u < P.n_slices && O.last_mb >= 0) {
            const VlcSliceOut N = slice_out[P.first_slice + sidx + 1u];
            if (N.last_mb >= 0 && N.first_rec / bpm <= (uint32_t)O.last_mb)
                atomicCAS(errors + blockIdx.x, 0u, (uint32_t)VLC_ERR_OVERLAP | ((uint32_t)slices[P.first_slice + sidx + 1u].code << 8));
        }
        for (uint32_t b0 = (uint32_t)lane; b0 < n_blocks; b0 += 256) {
            uint32_t r1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r1[u] = b0 + 64u * u < n_blocks ? rec[(b0 + 64u * u) * kVlcRecWords + 1] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t gid = r1[u] & 0xfffffu, k = r1[u] >> 24;
                if (b0 + 64u * u < n_blocks && gid < (uint32_t)ng && k <= 64u) rec[(b0 + 64u * u) * kVlcRecWords + 3] = atomicAdd(cnt + gid, k);
            }
        }
    }
    __syncthreads();
    // exclusive scan of the counters, in place: a thread sums `per` neighbours, the 256 sums are scanned by shuffles
    const int per = (ng + kVlcIndexThreads - 1) / kVlcIndexThreads;
    const int lo = min(tid * per, ng), hi = min(lo + per, ng);
    uint32_t sum = 0;
    for (int g = lo; g < hi; g++) sum += cnt[g];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wv; w++) run += part[w];
    for (int g = lo; g < hi; g++) {
        const uint32_t n = cnt[g];
        cnt[g] = run;
        run += n;
    }
    __syncthreads();
    for (int g = tid; g < ng; g += kVlcIndexThreads) P.grp_off[g] = cnt[g];
    if (tid == kVlcIndexThreads - 1) P.grp_off[ng] = run;
    // the maps, four macroblocks of a lane in flight
    const int mbs = G.mbw * G.mbh;
    const uint4* const mbrec = reinterpret_cast<const uint4*>(P.zbase);
    uint8_t* const m_qscale = reinterpret_cast<uint8_t*>(P.maps + G.off_qscale);
    uint8_t* const m_intra = reinterpret_cast<uint8_t*>(P.maps + G.off_intra);
    uint8_t* const m_repadd = reinterpret_cast<uint8_t*>(P.maps + G.off_repadd);
    uint8_t* const m_mb_dir = reinterpret_cast<uint8_t*>(P.maps + G.off_mb_dir);
    uint32_t* const m_mv_fwd = reinterpret_cast<uint32_t*>(P.maps + G.off_mv_fwd);
    uint32_t* const m_mv_bwd = reinterpret_cast<uint32_t*>(P.maps + G.off_mv_bwd);
    for (int mb0 = tid; mb0 < mbs; mb0 += 4 * kVlcIndexThreads) {
        uint4 m[4];
#pragma unroll
        for (int u = 0; u < 4; u++) m[u] = mb0 + u * kVlcIndexThreads < mbs ? mbrec[mb0 + u * kVlcIndexThreads] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int mb = mb0 + u * kVlcIndexThreads;
            if (mb >= mbs) break;
            m_qscale[mb] = (uint8_t)m[u].x;
            m_intra[mb] = (m[u].x & 0x100u) ? 255 : 0;
            if (P.type != 1) {
                m_repadd[mb] = (m[u].x & 0x200u) ? 255 : 0;
                m_mv_fwd[mb] = m[u].y;
            }
            if (P.type == 3) {
                m_mb_dir[mb] = (uint8_t)((m[u].x >> 10) & 3u);
                m_mv_bwd[mb] = m[u].z;
            }
        }
    }
}

// The parallel half: the symbols of every coded block, a lane per block.  One WAVE per slice walks the slice's records 64
// at a time; a lane reads its block's bits straight from memory (three dwords cover most blocks), decodes run / level
// pairs until the end-of-block code and writes them as entries (tile offset << 16 | level) to grp_off[group] + its place.
// Same symbol reading as leon_vlc.cpp's decode_block; a block that does not read the way k_vlc_parse counted it (it
// cannot, both read the same bits with the same tables) sets the picture's error word instead of leaving entries undefined.
__global__ __launch_bounds__(256) void k_vlc_blocks(const VlcSlice* __restrict__ slices, const VlcSliceOut* __restrict__ slice_out, int n_slices,
                                                    const VlcPic* __restrict__ pics, uint32_t* __restrict__ errors, VlcGeom G,
                                                    const VlcTables* __restrict__ T)
{
    __shared__ uint16_t s_fast[4096], s_long[512], s_zz[64];
    for (int i = threadIdx.x; i < 4096; i += 256) s_fast[i] = T->fast12[i];
    for (int i = threadIdx.x; i < 512; i += 256) s_long[i] = T->long9[i];
    if (threadIdx.x < 64) s_zz[threadIdx.x] = T->zz_off[threadIdx.x];
    __syncthreads();
    const int wv = (int)(threadIdx.x >> 6);
    const int j = blockIdx.x * 4 + wv;                               // one wave per slice
    const int lane = threadIdx.x & 63;
    if (j >= n_slices) return;
    const VlcSlice S = slices[j];
    const VlcPic P = pics[S.pic];
    const VlcSliceOut O = slice_out[j];
    const uint32_t rec_cap = (uint32_t)(G.mbw * G.mbh * vlc_blocks_per_mb(G.alpha));
    const uint32_t n_blocks = O.first_rec <= O.end_rec && O.end_rec <= rec_cap ? O.end_rec - O.first_rec : 0u;
    const VLC_G uint32_t* const rec = (const VLC_G uint32_t*)P.recs + (size_t)O.first_rec * kVlcRecWords;
    const VLC_G uint32_t* const bytes = (const VLC_G uint32_t*)S.bytes;
    const VLC_G uint32_t* const grp_off = (const VLC_G uint32_t*)P.grp_off;
    VLC_G uint32_t* const entries = (VLC_G uint32_t*)P.entries;
    uint32_t err = 0;
    for (uint32_t base = 0; base < n_blocks; base += 64) {
        const uint32_t b = base + (uint32_t)lane;
        if (b >= n_blocks) continue;
        const vlc_u4 rc = reinterpret_cast<const VLC_G vlc_u4*>(rec)[b];
        const uint32_t bit = rc.x, r1 = rc.y, dcw = rc.z;
        const uint32_t gid = r1 & 0xfffffu, bq = (r1 >> 20) & 7u, k = r1 >> 24;
        const bool intra = (r1 >> 23) & 1u;
        if (k == 0u || k > 64u || gid >= (uint32_t)G.n_groups) continue;
        const uint32_t at = grp_off[gid] + rc.w;              // k_vlc_index: the group's list, this block's place in it
        if (at + k > P.entries_cap || at + k < at) { err = VLC_ERR_SCRATCH; continue; }
        VLC_G uint32_t* out = entries + at;
        const uint32_t bbase = (bq * 16u) << 16;
        // the lane's window: 64 bits from `bit` on, refilled a dword at a time
        uint32_t next = bit >> 5;
        auto dword = [&](uint32_t i) -> uint32_t { return i < S.n_dwords ? __builtin_bswap32(bytes[i]) : 0u; };
        const uint32_t lead = bit & 31u;
        uint64_t w = (((uint64_t)dword(next) << 32) | dword(next + 1u)) << lead;
        int avail = 64 - (int)lead;
        next += 2u;
        uint32_t written = 0u;
        int n = 0;
        if (intra) {
            if (dcw & 0x10000u) out[written++] = bbase | (dcw & 0xffffu);
            n = 1;
        } else if ((uint32_t)(w >> 62) & 2u) {                       // '1s': run 0, level +-1 in first position
            out[written++] = bbase | (uint32_t)(uint16_t)(int16_t)(((uint32_t)(w >> 62) & 1u) ? -1 : 1);
            w <<= 2; avail -= 2;
            n = 1;
        }
        bool ok = false;
        for (int it = 0; it < 66; it++) {
            if (avail <= 32) { w |= (uint64_t)dword(next++) << (32 - avail); avail += 32; }
            const uint32_t f = s_fast[(uint32_t)(w >> 52)];
            const int flen = (int)(f & 0xfu);
            int run_len, level, used;
            if (flen) {
                if (f & 0x10u) { ok = true; break; }                 // end of block
                used = flen;
                run_len = (int)((f >> 5) & 31u);
                level = (int)(int16_t)(uint16_t)f >> 10;
            } else if ((w >> 58) == 1u) {                            // escape: 6-bit run, 8- or 16-bit level
                run_len = (int)((w >> 52) & 63);
                level = (int)((w >> 44) & 255);
                used = 20;
                if (level == 0) { level = (int)((w >> 36) & 255); used = 28; }
                else if (level == 128) { level = (int)((w >> 36) & 255) - 256; used = 28; }
                else if (level > 128) level -= 256;
            } else {
                const uint32_t e = (w >> 57) == 0u ? s_long[(uint32_t)(w >> 48) & 511u] : 0u;
                if (e == 0u) break;
                const int len = (int)(e & 31u);
                run_len = (int)((e >> 5) & 31u);
                level = (int)(e >> 10);
                if ((w >> (63 - len)) & 1) level = -level;
                used = len + 1;
            }
            w <<= used; avail -= used;
            n += run_len;
            if (n > 63 || written >= k) break;
            // (a level of 0 -- an escape that codes nothing -- is written too: k_vlc_parse counted the symbol, and an entry
            // with level 0 puts a zero where zero is)
            out[written++] = bbase | ((uint32_t)s_zz[n] << 16) | (uint32_t)(uint16_t)(int16_t)level;
            n++;
        }
        if (!ok || written != k) {
            err = VLC_ERR_COEF;
            for (; written < k; written++) out[written] = 0u;        // nothing undefined in the lists
        }
    }
    if (err) atomicCAS(errors + S.pic, 0u, err | ((uint32_t)S.code << 8));
}

}  // namespace leon
