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
//   k_vlc_parse    one lane per slice: macroblock maps straight into the picture's arrays; coefficients as
//                  block records (header {group id, count} + entries) into the slice's scratch strip, the
//                  count added to the group's counter.
//   k_vlc_offsets  one workgroup per picture: exclusive scan of the group counters -> grp_off (what
//                  leon_sparse_picture wants), counters back to zero (they become cursors).
//   k_vlc_gather   one lane per slice again: every block record moves to grp_off[group] + cursor (atomic add
//                  of the block's count).  Entries of a group are "in no particular order" (include/leon_vlc.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace leon {

struct VlcTables {               // device copy of leon_vlc_gpu_tables, LDS part first
    uint32_t fast12[4096];
    int32_t motion_s[2048];
    int32_t mba[2048];
    int32_t cbp[512];
    int32_t mbtype[4][64];
    int32_t dc_lum[128], dc_chr[256];
    uint32_t zz_off[64];
    int32_t coef16[65536];       // stays in global memory: long codes and escapes only
};
static constexpr int kVlcLdsWords = 4096 + 2048 + 2048 + 512 + 256 + 128 + 256 + 64;

struct VlcGeom {
    int32_t mbw, mbh, gy, gc, n_y, n_c, n_groups, alpha;
};

struct VlcPic {                  // one picture of the window
    int32_t type, full_pel_fwd, fwd_rsize, full_pel_bwd, bwd_rsize, pad;
    uint8_t* qscale;
    uint8_t* intra;
    uint8_t* repadd;
    uint8_t* mb_dir;
    int16_t* mv_fwd;
    int16_t* mv_bwd;
    uint32_t* grp_cnt;           // [n_groups + 1], zero on entry
    uint32_t* grp_off;           // [n_groups + 1]
    uint32_t* entries;
    uint32_t entries_cap;
    uint32_t pad2;
    uint32_t* error;             // one word, zero on entry; first error code of any of its slices
};

struct VlcSlice {
    const uint32_t* bytes;       // the GOP's stream copy (4-byte aligned, zero padded)
    uint32_t n_dwords;           // readable dwords of it
    uint32_t n_bytes;            // its real length
    uint32_t bit_pos;            // first bit behind the slice start code
    uint32_t end_byte;           // first byte behind the slice
    int32_t code;                // slice_vertical_position
    uint32_t pic;
    uint32_t scratch_cap;        // words
    uint32_t pad;
    uint32_t* scratch;
};

enum { VLC_ERR_MBA = 1, VLC_ERR_ADDR, VLC_ERR_TYPE, VLC_ERR_MOTION, VLC_ERR_CBP, VLC_ERR_COEF, VLC_ERR_INDEX, VLC_ERR_END, VLC_ERR_DC,
       VLC_ERR_SCRATCH };

struct VlcWin {
    const uint32_t* base;
    uint32_t next, nd;           // next dword to load, dwords there are
    uint64_t w;                  // the stream from `pos` on, left aligned, `avail` bits valid, zeros below
    int avail;
    uint32_t pos;                // in bits, from base
    __device__ __forceinline__ void init(const uint32_t* b, uint32_t n_dwords, uint32_t bit_pos)
    {
        base = b; nd = n_dwords; pos = bit_pos;
        next = bit_pos >> 5;
        const uint32_t lead = bit_pos & 31u;
        const uint32_t d0 = load(), d1 = load();
        w = (((uint64_t)d0 << 32) | d1) << lead;
        avail = 64 - (int)lead;
    }
    __device__ __forceinline__ uint32_t load()
    {
        const uint32_t d = next < nd ? __builtin_bswap32(__builtin_nontemporal_load(base + next)) : 0u;
        next++;
        return d;
    }
    // at least 32 valid bits afterwards (one symbol of the syntax takes at most 28)
    __device__ __forceinline__ void fill()
    {
        if (avail <= 32) {
            w |= (uint64_t)load() << (32 - avail);
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
    uint32_t fast12[4096];
    int32_t motion_s[2048];
    int32_t mba[2048];
    int32_t cbp[512];
    int32_t mbtype[4][64];
    int32_t dc_lum[128], dc_chr[256];
    uint32_t zz_off[64];
};
static_assert(sizeof(VlcLds) == kVlcLdsWords * 4, "LDS copy and VlcTables disagree");

struct VlcCtx {                  // per lane: the state a slice carries from macroblock to macroblock
    int mb_addr, mb_row, mb_col, rc_addr;
    int fw_h, fw_v, fw_h_prev, fw_v_prev, bw_h, bw_v, bw_h_prev, bw_v_prev, prev_dir;
    int dc_y, dc_cr, dc_cb, dc_a, qs;
    int mb_intra;
    uint32_t* out;               // next free word of the scratch strip
    uint32_t* out_end;
};

// decoders/jsv.js:831-893, as motion_component of leon_vlc.cpp
__device__ __forceinline__ int vlc_motion_component(VlcWin& r, const VlcLds& L, int prev, int rsize, int f, int& err)
{
    r.fill();
    const int32_t e = L.motion_s[r.peek(11)];
    if (e == 0) { err = VLC_ERR_MOTION; return prev; }
    r.drop(e >> 16);
    const int code = (e & 0xffff) - 16;
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
__device__ __forceinline__ int vlc_block(VlcWin& r, const VlcLds& L, const VlcTables* __restrict__ T, const VlcGeom& G, const VlcPic& P,
                                         VlcCtx& c, int block)
{
    uint32_t gid, bq;
    if (block < 4 || block >= 6) {                           // luma, or the A component (blocks 6..9, placed like luma)
        const int lb = block < 4 ? block : block - 6;
        const int qb = c.mb_col * 2 + (lb & 1);
        gid = (uint32_t)((2 * c.mb_row + (lb >> 1)) * G.gy + (qb >> 3)) + (block < 4 ? 0u : (uint32_t)(G.n_y + 2 * G.n_c));
        bq = (uint32_t)(qb & 7);
    } else {
        gid = (uint32_t)(G.n_y + (block == 5 ? G.n_c : 0) + c.mb_row * G.gc + (c.mb_col >> 3));
        bq = (uint32_t)(c.mb_col & 7);
    }
    const uint32_t boff = (bq * 16u) << 16;
    if (c.out + 65 > c.out_end) return VLC_ERR_SCRATCH;      // a block record: header + at most 64 entries
    uint32_t* const rec = c.out;
    int k = 0, n = 0;
    if (c.mb_intra) {
        r.fill();
        const int32_t e = block < 4 || block >= 6 ? L.dc_lum[r.peek(7)] : L.dc_chr[r.peek(8)];
        if (e == 0) return VLC_ERR_DC;
        r.drop(e >> 16);
        const int size = e & 0xffff;
        int predictor = block < 4 ? c.dc_y : block >= 6 ? c.dc_a : block == 4 ? c.dc_cr : c.dc_cb;
        int dc = predictor;
        if (size > 0) {
            const int differential = (int)r.get(size);
            dc = (differential & (1 << (size - 1))) ? predictor + differential
                                                    : predictor + ((int)(0xffffffffu << size) | (differential + 1));
        }
        if (block < 4) c.dc_y = dc; else if (block >= 6) c.dc_a = dc; else if (block == 4) c.dc_cr = dc; else c.dc_cb = dc;
        if ((int16_t)dc != 0) rec[1 + k++] = boff | (uint16_t)(int16_t)dc;
        n = 1;
    }
    bool first = n == 0;
    for (;;) {
        r.fill();
        const uint32_t p12 = (uint32_t)(r.w >> 52);
        uint32_t f = L.fast12[p12];
        // the first symbol of a block: '1s' is run 0, level +-1, and there is no end-of-block code
        if (first && (p12 >> 11)) f = ((uint32_t)(uint16_t)(int16_t)(((p12 >> 10) & 1u) ? -1 : 1) << 16) | 2u;
        first = false;
        const int flen = (int)(f & 0x7fu);
        int run_len, level;
        if (flen) {
            r.drop(flen);
            if (f & 0x80u) break;                             // end of block
            run_len = (int)((f >> 8) & 0xffu);
            level = (int)(int16_t)(f >> 16);
        } else {
            // longer codes and escapes
            const uint64_t w = r.w;
            const int32_t e = T->coef16[w >> 48];
            if (e == 0) return VLC_ERR_COEF;
            const int len = e >> 16, coeff = e & 0xffff;
            int used;
            if (coeff == 0xffff) {                           // escape: 6-bit run, 8- or 16-bit level
                run_len = (int)((w >> 52) & 63);
                level = (int)((w >> 44) & 255);
                used = 20;
                if (level == 0) { level = (int)((w >> 36) & 255); used = 28; }
                else if (level == 128) { level = (int)((w >> 36) & 255) - 256; used = 28; }
                else if (level > 128) level -= 256;
            } else {
                run_len = coeff >> 8;
                level = coeff & 0xff;
                if ((w >> (63 - len)) & 1) level = -level;
                used = len + 1;
            }
            r.drop(used);
        }
        n += run_len;
        if (n > 63) return VLC_ERR_INDEX;
        const uint32_t zo = L.zz_off[n++];
        if (level != 0) rec[1 + k++] = boff | (zo << 16) | (uint16_t)(int16_t)level;
        if (r.pos > (uint32_t)(r.nd << 5)) return VLC_ERR_END;
    }
    if (k) {
        rec[0] = (gid << 7) | (uint32_t)k;
        c.out = rec + 1 + k;
        atomicAdd(P.grp_cnt + gid, (uint32_t)k);
    }
    return 0;
}

// decoders/jsv.js:725-828 (+ B pictures), as decode_macroblock of leon_vlc.cpp.  0 = macroblock read, 1 = stop
// silently (an address past the picture), > 1 an error
__device__ __forceinline__ int vlc_macroblock(VlcWin& r, const VlcLds& L, const VlcTables* __restrict__ T, const VlcGeom& G, const VlcPic& P,
                                              VlcCtx& c, bool& slice_begin)
{
    const int type = P.type, mbsize = G.mbw * G.mbh;
    int increment = 0, t;
    auto mba = [&]() -> int {
        r.fill();
        const int32_t e = L.mba[r.peek(11)];
        if (e == 0) return -1;
        r.drop(e >> 16);
        return e & 0xffff;
    };
    t = mba();
    while (t == 34) t = mba();                               // stuffing
    while (t == 35) { increment += 33; t = mba(); }          // escape
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
            if (type != 1) *reinterpret_cast<uint32_t*>(P.mv_fwd + 2 * a) = (uint32_t)(uint16_t)(int16_t)c.fw_h | ((uint32_t)(uint16_t)(int16_t)c.fw_v << 16);
            if (type == 3) {
                *reinterpret_cast<uint32_t*>(P.mv_bwd + 2 * a) = (uint32_t)(uint16_t)(int16_t)c.bw_h | ((uint32_t)(uint16_t)(int16_t)c.bw_v << 16);
                P.mb_dir[a] = (uint8_t)c.prev_dir;
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
    r.fill();
    const int32_t te = L.mbtype[type][r.peek(6)];
    if (te == 0) return VLC_ERR_TYPE + 1;
    r.drop(te >> 16);
    const int mb_type = te & 0xffff;
    c.mb_intra = mb_type & 0x01;
    const int mot_fw = mb_type & 0x08, mot_bw = mb_type & 0x04;
    if (mb_type & 0x10) c.qs = (int)r.get(5);
    P.qscale[mb] = (uint8_t)c.qs;
    P.intra[mb] = c.mb_intra ? 255 : 0;
    if (c.mb_intra) {
        c.fw_h = c.fw_h_prev = 0; c.fw_v = c.fw_v_prev = 0;
        c.bw_h = c.bw_h_prev = 0; c.bw_v = c.bw_v_prev = 0;
        c.prev_dir = 0;
        if (type != 1) P.repadd[mb] = 255;                   // jsv.js:1502-1505
    } else {
        c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
        int err = 0;
        if (mot_fw) {
            c.fw_h_prev = vlc_motion_component(r, L, c.fw_h_prev, P.fwd_rsize, 1 << P.fwd_rsize, err);
            c.fw_h = P.full_pel_fwd ? c.fw_h_prev * 2 : c.fw_h_prev;
            c.fw_v_prev = vlc_motion_component(r, L, c.fw_v_prev, P.fwd_rsize, 1 << P.fwd_rsize, err);
            c.fw_v = P.full_pel_fwd ? c.fw_v_prev * 2 : c.fw_v_prev;
        } else if (type == 2) {
            c.fw_h = c.fw_h_prev = 0;
            c.fw_v = c.fw_v_prev = 0;
        }
        if (mot_bw) {
            c.bw_h_prev = vlc_motion_component(r, L, c.bw_h_prev, P.bwd_rsize, 1 << P.bwd_rsize, err);
            c.bw_h = P.full_pel_bwd ? c.bw_h_prev * 2 : c.bw_h_prev;
            c.bw_v_prev = vlc_motion_component(r, L, c.bw_v_prev, P.bwd_rsize, 1 << P.bwd_rsize, err);
            c.bw_v = P.full_pel_bwd ? c.bw_v_prev * 2 : c.bw_v_prev;
        }
        if (err) return err + 1;
        if (type != 1) *reinterpret_cast<uint32_t*>(P.mv_fwd + 2 * mb) = (uint32_t)(uint16_t)(int16_t)c.fw_h | ((uint32_t)(uint16_t)(int16_t)c.fw_v << 16);
        if (type == 3) {
            *reinterpret_cast<uint32_t*>(P.mv_bwd + 2 * mb) = (uint32_t)(uint16_t)(int16_t)c.bw_h | ((uint32_t)(uint16_t)(int16_t)c.bw_v << 16);
            c.prev_dir = (mot_fw ? 1 : 0) | (mot_bw ? 2 : 0);
            P.mb_dir[mb] = (uint8_t)c.prev_dir;
        }
    }
    int cbp = 0;
    if (mb_type & 0x02) {
        r.fill();
        const int32_t ce = L.cbp[r.peek(9)];
        if (ce == 0) return VLC_ERR_CBP + 1;
        r.drop(ce >> 16);
        cbp = ce & 0xffff;
    } else if (c.mb_intra) cbp = 0x3f;
    int apat = 0;
    if (G.alpha) apat = c.mb_intra ? 0xf : (int)r.get(4);
    for (int block = 0, mask = 0x20; block < 6; block++, mask >>= 1)
        if (cbp & mask) { const int e = vlc_block(r, L, T, G, P, c, block); if (e) return e + 1; }
    for (int block = 6, mask = 0x8; block < 10; block++, mask >>= 1)
        if (apat & mask) { const int e = vlc_block(r, L, T, G, P, c, block); if (e) return e + 1; }
    return 0;
}

// LDS (36.75 KB per workgroup) allows four waves per SIMD: let the registers go that far too (128 VGPRs, no spills)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_vlc_parse(const VlcSlice* __restrict__ slices, uint32_t* __restrict__ slice_words, int n_slices,
                                                   const VlcPic* __restrict__ pics, VlcGeom G, const VlcTables* __restrict__ T)
{
    __shared__ VlcLds L;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(T);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&L);
        for (int i = threadIdx.x; i < kVlcLdsWords; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_slices) return;
    const VlcSlice S = slices[j];
    const VlcPic P = pics[S.pic];
    VlcWin r;
    r.init(S.bytes, S.n_dwords, S.bit_pos);
    VlcCtx c{};
    c.mb_addr = (S.code - 1) * G.mbw - 1;                     // decoders/jsv.js:683-706
    c.mb_row = S.code - 1;
    c.mb_col = -1;
    c.rc_addr = c.mb_addr;
    c.dc_y = c.dc_cr = c.dc_cb = c.dc_a = 128;
    c.out = S.scratch;
    c.out_end = S.scratch + S.scratch_cap;
    c.qs = (int)r.get(5);
    while (r.get(1) && r.pos < (S.end_byte << 3)) r.get(8);
    bool slice_begin = true;
    int err = 0;
    for (;;) {
        const int rc = vlc_macroblock(r, L, T, G, P, c, slice_begin);       // 1: the reference's silent return, the loop goes on
        if (rc > 1) { err = rc - 1; break; }
        // next_bits_are_start_code (decoders/jsv.js:1710-1760): byte aligned 00 00 01, or the end of the data
        r.fill();
        const uint32_t i = (r.pos + 7u) >> 3, skip = (0u - r.pos) & 7u;
        if (i + 2u >= S.n_bytes) break;
        if ((uint32_t)((r.w << skip) >> 40) == 1u) break;
        if (i >= S.end_byte) { err = VLC_ERR_END; break; }    // behind the start code the host found: ran over it
    }
    slice_words[j] = (uint32_t)(c.out - S.scratch);
    if (err) atomicCAS(P.error, 0u, (uint32_t)err | ((uint32_t)S.code << 8));
}

// exclusive scan of a picture's group counters; the counters go back to zero (k_vlc_gather's cursors)
__global__ __launch_bounds__(256) void k_vlc_offsets(const VlcPic* __restrict__ pics, VlcGeom G)
{
    __shared__ uint32_t part[256];
    const VlcPic P = pics[blockIdx.x];
    const int ng = G.n_groups, per = (ng + 255) / 256;
    const int lo = min((int)threadIdx.x * per, ng), hi = min(lo + per, ng);
    uint32_t sum = 0;
    for (int g = lo; g < hi; g++) sum += P.grp_cnt[g];
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
        const uint32_t n = P.grp_cnt[g];
        P.grp_off[g] = run;
        P.grp_cnt[g] = 0;
        run += n;
    }
    if (threadIdx.x == 255) P.grp_off[ng] = part[255];
}

__global__ __launch_bounds__(256) void k_vlc_gather(const VlcSlice* __restrict__ slices, const uint32_t* __restrict__ slice_words, int n_slices,
                                                    const VlcPic* __restrict__ pics)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_slices) return;
    const VlcSlice S = slices[j];
    const VlcPic P = pics[S.pic];
    const uint32_t* rec = S.scratch;
    const uint32_t* const end = rec + min(slice_words[j], S.scratch_cap);
    while (rec < end) {
        const uint32_t h = *rec++;
        const uint32_t gid = h >> 7, k = h & 127u;
        const uint32_t at = P.grp_off[gid] + atomicAdd(P.grp_cnt + gid, k);
        if (k > 64u || rec + k > end || at + k > P.entries_cap) break;        // cannot happen with k_vlc_parse's records
        for (uint32_t i = 0; i < k; i++) P.entries[at + i] = rec[i];
        rec += k;
    }
}

}  // namespace leon
