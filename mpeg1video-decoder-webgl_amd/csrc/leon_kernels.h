// leon_kernels.h -- CDNA4 (gfx950) device code of the macroblock-reconstruction path.
//
// Three kernels, all HBM-bound integer/byte work (no MFMA: there is no dense
// contraction in this path):
//   k_recon      dequant + 8x8 IDCT (two passes with the reference's x0.4 int16
//                hand-off) + forward/backward/bidirectional half-pel motion
//                compensation + residual add + clamp, fused, for a BATCH of
//                mutually independent pictures.            (K1a+K1b+K2+K2-B)
//   k_rgba_*     YCbCr 4:2:0 -> RGBA8, fp64 "CPU twin" or fp32 "GL" flavour. (K3)
//   k_copy16     streaming 16 B/lane copy = the measured HBM roofline.
//
// What the arithmetic follows in /root/reference:
//   dequant + column pass   decoders/shaders/mpeg1video.js:19-24 (COL_INT_3, COL_INT_5)
//   row pass + MC + clamp   decoders/shaders/mpeg1video.js:24-29 (ROWSCOM_INT4, INTER_INT1)
//   predictor arithmetic    decoders/jsv.js:895-1129 (copyMacroblock), texel-granular
//                           CLAMP_TO_EDGE of jsv.js:216-217
//   colour conversion       player/easybits.player.js:2674-2785, player/parts/end.js:77-156
//
// Work decomposition of k_recon: one 64-lane wave = one task = TWO "block groups" that share
// their macroblocks (luma: the upper and lower 64x8 halves of four macroblocks; chroma: the Cb
// and the Cr group of one block row); a group = 8 horizontally adjacent 8x8 blocks.  Waves never
// talk to each other (no __syncthreads); each owns a private LDS strip.  One kernel instance
// per picture type and boundary format (k_recon<type, sparse>).
//   prologue every load that does not depend on the macroblock maps is requested first (first
//            coefficient rows or entry runs, matrix columns), then the maps and vectors; when
//            those arrive, the reference rows of BOTH halves (one aligned 12-byte row per lane)
//   stage 1  lane (r,b): coefficient row r of block b -> LDS tile [r][b][c]
//            (sparse boundary: the tile is cleared and the group's entries are scattered into it)
//   stage 2  lane (c = lane>>3, b = lane&7): column c of block b: 8 LDS reads, branch-free dequant
//            of the rows that are live anywhere in the wave (scalar liveness masks; zeros stay
//            zero like the shader's `if (X == 0.) continue`), butterfly, floor(v*0.4f), int16
//            hand-off wrap on a rare path, trunc(5w/2), written transposed to LDS as int32
//   stage 3  lane (n = lane>>3, b = lane&7): row n of block b: two 16-byte LDS reads, butterfly
//   stage 4  same lanes: the 8 lanes of a group hold 64 contiguous samples of ONE picture row, so
//            reference fetches and stores coalesce; lower reference row from lane+8; 8 predicted
//            samples via v_alignbyte + v_lerp_u8, residual add, v_ashr_pk_u8_i32, one 8-byte store
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef LEON_CARRY
#define LEON_CARRY 1
#endif
#ifndef LEON_PAIR_B
#define LEON_PAIR_B 1
#endif

namespace leon {

struct PicDesc {                 // one per picture of a batch, device resident
    const int16_t* coef[4];      // T1: Y, Cb, Cr raw levels (+ A of a yuva stream)
    const uint8_t* qscale;       // T2
    const uint8_t* intra;        // T3
    const uint8_t* repadd;       // T4
    const uint8_t* mb_dir;       // B only
    const int16_t* mv_fwd;       // T5
    const int16_t* mv_bwd;       // B only
    uint8_t*       out;          // slot base: [Y | Cb | Cr]
    const uint8_t* ref_fwd;
    const uint8_t* ref_bwd;
    int32_t        type;         // 1 I, 2 P, 3 B
    uint32_t       n_entries;    // sparse boundary: entries[] length
    const uint32_t* grp_off;     // sparse boundary (include/leon_vlc.h): prefix offsets per 64x8 group
    const uint32_t* entries;     // (tile byte offset << 16) | level
    uint8_t*       rgba;         // fused display conversion: frame_w x frame_h RGBA8 destination (else null)
    int32_t        no_planes;    // with rgba: the slot's planes are not written (picture is never a reference)
    int32_t        pad_;
    const struct QTables* qt;    // the quantiser matrices of the picture's SEQUENCE (the reference reloads them at every sequence
                                 // header, decoders/jsv.js:540-558): one of the decoder's matrix sets
};

struct Geom {
    int32_t cw, ch;              // coded luma size
    int32_t mbw, mbh;
    int32_t gY, gC;              // block groups per block row (luma, chroma)
    int32_t tasksY, tasksC;      // gY*mbh luma tasks (two block rows each), gC*mbh chroma tasks (Cb + Cr)
    int32_t tasks_per_pic;       // tasksY + tasksC
    int32_t wg_per_pic;          // ceil(tasks_per_pic / 4): a workgroup never straddles pictures
    int32_t n_pics;
    int32_t n_wg;                // grid size
    uint32_t inv_wg_per_pic;     // ceil(2^32 / wg_per_pic)
    uint32_t inv_gY, inv_gC;     // ceil(2^32 / gY), ceil(2^32 / gC)
    int32_t fw, fh;              // display crop (fused display conversion)
    int32_t alpha;               // yuva: a fourth, luma-sized plane behind Cr; its tasks follow the chroma tasks
};

struct QTables {                 // T6, per sequence: 256 bytes = what a wave stages in its LDS strip (stage_tables)
    uint8_t qmT[2][8][8];        // [0 intra | 1 non-intra][column c][row i] = Q[i][c]
    uint8_t pmT[8][8];           // premultiplier, [column c][row i]
    uint8_t pad[64];             // (in LDS these 64 bytes carry a display task's vectors: kOffCarry)
};
struct Tables {                  // per decoder
    QTables q;                   // matrix set 0: leon_set_quant_matrices
    int32_t rgba_lut[1280];      // fused display conversion: fixed-point terms of YCbCrToRGBA (leon_rgba_lut.h)
};

static constexpr int kWavesPerWG = 4;
// Launch block sizes and the kernels' __launch_bounds__ come from the same constants: a launch with
// more threads than the bound fails at launch time ("unspecified launch failure" from
// hipGetLastError -- what a 512-thread block-size sweep of the RGBA kernels ran into in round 1).
static constexpr int kReconMaxThreads = 384;
static constexpr int kWavesPerWGPairB = 4;       // (6 waves per workgroup for the dense B display kernel were tried: launch_recon_type)
static constexpr int kRgbaBlock = 256;
// A wave's LDS strip.  The tile holds BOTH block groups of a task, [half][row][block][column] int16, and is
// worked on in place: coefficients -> (column pass) the int16 hand-off values w -> (row pass reads them).
static constexpr int kLdsHalf = 1024;            // 8 rows x 128 B of int16: one block group
static constexpr int kLdsTile = 2 * kLdsHalf;
static constexpr int kLdsQtab = 256;             // a QTables: both matrices and the premultiplier (192 B) + 64 spare
static constexpr int kLdsSlots = 128;            // column pass: the ids of the live columns, one byte each
static constexpr int kOffQtab = kLdsTile;
static constexpr int kOffCarry = kOffQtab + 192;  // display kernels: the 64 bytes of the table strip no table uses hold the task's 8 + 8 vectors
static constexpr int kOffSlots = kLdsTile + kLdsQtab;
static constexpr int kLdsPerWave = kOffSlots + kLdsSlots;
// fused display conversion: the Y rows of a half change lanes through a 512-byte park (it shares its place with the
// column ids, which are dead by then), and the Cb and Cr samples of the task's 8 macroblocks (8 rows x 64 bytes
// each) wait in a stash for the luma parts of the same task
static constexpr int kLdsYpark = 512;
static constexpr int kOffYpark = kOffSlots;
static constexpr int kOffStash = kOffYpark + kLdsYpark;
static constexpr int kLdsStash = 2 * 512;
static constexpr int kLdsPerWaveDisplay = kOffStash + kLdsStash;      // 3840: 4 waves + the tables = 20 KB, 8 workgroups per CU
static constexpr int kLdsLut = 1280 * 4;         // display kernels: Tables::rgba_lut, one copy per workgroup, in front of the waves' strips
static constexpr int kLutShift = 21;             // = LEON_RGBA_LUT_SHIFT (static_assert in leon_hip.cpp)
static constexpr int kLdsPerWaveDisplayAlpha = kLdsPerWaveDisplay + 1024;   // yuva: + the parked A samples of a luma part
// dense P / B display tasks (round 4): the two luma parts of a task share ONE liveness scan and ONE column pass -- the right part's
// tile lies behind the wave's strip
// The wave's strip, two layouts:
//   0 (all kernels but the next): [tile 2048 | tables 192 + carried vectors 64 | column ids / Y park 512 | stash 1024 (| A park)]
//   1 (dense P / B display, two tiles): [tile L 2048 | carried vectors 64 | column ids 256 | stash 1024 | tile R 2048] = 5440 bytes:
//     4 waves + the conversion tables = 26.25 KB, SIX workgroups per CU (with layout 0 plus a second tile: 28 KB, five -- and the B
//     kernel loses 7 % at five).  What went: the wave's LDS copy of the quantiser tables (the column pass reads its two 8-byte columns
//     from the picture's QTables in memory: 256 bytes that every wave of the launch reads, L1 hits), and the Y park, which now lies in
//     the tile half whose row pass has just read it out.
template <int LAYOUT> struct Lay;
template <> struct Lay<0> { static constexpr int carry = kOffQtab + 192, slots = kOffSlots, stash = kOffStash; static constexpr bool tables_in_lds = true, park_in_tile = false; };
template <> struct Lay<1> { static constexpr int carry = kLdsTile, slots = kLdsTile + 64, stash = kLdsTile + 64 + 256, tile_r = kLdsTile + 64 + 256 + kLdsStash;
                            static constexpr bool tables_in_lds = false, park_in_tile = true; };
static constexpr int kLdsPerWaveDisplayPair = Lay<1>::tile_r + kLdsTile;      // 5440
static constexpr int kOffTileR = Lay<1>::tile_r;
#ifndef LEON_PAIR_LUMA
#define LEON_PAIR_LUMA 1
#endif
#ifndef LEON_NINTH_FROM_BELOW
#define LEON_NINTH_FROM_BELOW 1
#endif
#ifndef LEON_PRIO
#define LEON_PRIO 0        // experiment (round 4): wave priority by progress through a display task, all picture types (1: rising, 2: falling)
#endif
#ifndef LEON_PRIO_B
#define LEON_PRIO_B 3      // B display tasks: a wave starts at priority 3 and drops to 0 (2: after its chroma part, 3: when the luma parts' coefficient
                          // loads are out, 4: after their column pass, 5: before the second luma part) -- young waves first: their loads are on the way
                          // while the older ones compute.  One box, alternating, ms per mixed B launch: 1.027-1.031 without, 0.992-0.993 with 3 (4: the
                          // same, 5: 1.018, 2: 0.998-1.004); the step 5.83-5.85 -> 5.73-5.75 ms.  For ALL types (LEON_PRIO=2) I and P got slower.
#endif
#ifndef LEON_PRIO_P
#define LEON_PRIO_P 0
#endif
#define LEON_PRIO_OF(TYPE) (LEON_PRIO ? LEON_PRIO : ((TYPE) == 3 ? LEON_PRIO_B : (TYPE) == 2 ? LEON_PRIO_P : 0))      // (recon_luma_pair is the dense, non-alpha path only)
#ifndef LEON_RGBA_AUX
#define LEON_RGBA_AUX 2    // cache policy bits of the frames' stores (1 sc0, 2 nt, 16 sc1).  nt: the GPU never reads a frame again, and written
                          // through the caches like everything else it pushes the reference planes out -- round 4, one box, alternating: 5.944 ->
                          // 5.818 ms per step (I -5 %, P -3.6 %, mixed B -1.2 %); sc0 / sc1: nothing.  (Round 2 measured nt on ALL stores: -4 %,
                          // the planes are read again.)
#endif
#ifndef LEON_ABL
#define LEON_ABL 0      // ablation builds (tools/ab_build.sh x -DLEON_ABL=n): 1 no reference loads, 2 no RGBA stores, 4 no display conversion at all; WRONG output, timing only
#endif
#ifndef LEON_NO_PLANES_BRANCH
#define LEON_NO_PLANES_BRANCH 1
#endif

// ---- small helpers -----------------------------------------------------------

// Pointers read out of a PicDesc are generic; telling the compiler they are global
// turns flat_load/flat_store into global_load/global_store with an SGPR base.
#define LEON_GLOBAL __attribute__((address_space(1)))
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v3u __attribute__((ext_vector_type(3)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ const LEON_GLOBAL T* gptr(const T* p) { return (const LEON_GLOBAL T*)p; }
template <typename T>
__device__ __forceinline__ LEON_GLOBAL T* gptr_mut(T* p) { return (LEON_GLOBAL T*)p; }

// a*K + c with a 24-bit multiplier, one full-rate VALU op.  (hipcc turns __mul24 by
// a constant into the quarter-rate v_mul_lo_u32 once it has proven the operand small.)
__device__ __forceinline__ int mad24k(int a, int k, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}

// {sat_u8(a >> SH), sat_u8(b >> SH), sat_u8(c >> SH), sat_u8(d >> SH)}: two v_ashr_pk_u8_i32 (gfx950), the second one
// into the high half of the same register (op_sel[3]; each leaves the other half alone).  Through asm: hipcc's own
// pattern match of med3(ashr)|shl onto this instruction (ROCm 7.2) forgets that it leaves half of the destination
// alone, and the builtin does not know the half-selecting form.
template <int SH>
__device__ __forceinline__ uint32_t sat_pk4(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "n"(SH));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d), "n"(SH));
    return r;
}

// buffer resource over [p, p + 2 GiB): wave-uniform base in SGPRs, 32-bit byte offsets per lane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
// Lanes that must not touch memory carry this bit in their 32-bit buffer offset: it is past
// num_records of every resource above, so the load returns 0 and the store is dropped by the
// bounds check -- no exec-mask branch around the instruction, and the number of outstanding
// memory operations stays a compile-time constant (precise s_waitcnt counts instead of 0).
constexpr uint32_t kOobBit = 0x80000000u;
// Cache policy of data that is read exactly once (coefficient rows, entry lists): the `nt` bit.
// The reference rows are re-read by neighbouring groups and must keep their place in L1/L2;
// measured +3 % on the bench workload (sc0 / sc1 on the same loads: nothing; nt on the stores: -4 %).
constexpr int kAuxStreamOnce = 2;
// The dense boundary's coefficient rows go from memory straight into the wave's LDS tile (gfx950:
// buffer_load_dwordx4 ... lds; lane i's 16 bytes land at tile + 16 i, the layout stage 1 reads) -- no registers
// in between (four VGPRs held across half a task cost the B path a wave of occupancy) and no ds_write.  The
// compiler does not order LDS reads behind such a load; the waits are explicit (wait_vmem_all) and sit where
// everything outstanding has long arrived: in front of the reference fetches of a task and in front of the
// stores of a half.  Lanes that must not load (kOobBit) get zeros.
__device__ __forceinline__ void coef_rows_to_lds(const void* plane, char* tile, uint32_t voff, uint32_t soff)
{
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(plane);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)tile, 16, (int)voff, (int)soff, 0, kAuxStreamOnce);
}
// every vector memory operation of this wave has completed (loads, LDS-direct loads, stores)
__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lds_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// lanes whose value is non-zero, as a scalar mask: ONE v_cmp (the ballot builtin on a
// 16-bit-derived compare costs three vector instructions with this compiler)
__device__ __forceinline__ uint64_t lanes_nonzero(int v)
{
    uint64_t m;
    asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(m) : "v"(v));
    return m;
}

// typed load at base + 32-bit byte offset: global_load with an SGPR base and a VGPR offset
template <typename T>
__device__ __forceinline__ T ldg(const LEON_GLOBAL void* base, uint32_t off)
{
    return *(const LEON_GLOBAL T*)((const LEON_GLOBAL char*)base + off);
}

// sign(v) in {-1, 0, 1}
__device__ __forceinline__ int sign3(int v)
{
    int d;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(d) : "v"(v));
    return d;
}
// opaque 24-bit multiply (keeps hipcc from re-deriving 24-bit operand tricks around it)
__device__ __forceinline__ int mul24_asm(int a, int b)
{
    int d;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// GLSL int '/' 256 (truncation toward zero)
__device__ __forceinline__ int div256(int t)
{
    return (int)(__umul24((unsigned)t >> 31, 255u) + (unsigned)t) >> 8;
}

__device__ __forceinline__ int med3i(int v, int lo, int hi) { return min(max(v, lo), hi); }

// n / d by multiply-high with inv = ceil(2^32/d) (0 encodes d == 1); exact for n*d < 2^32
__device__ __forceinline__ int div_inv(int n, uint32_t inv) { return inv ? (int)__umulhi((uint32_t)n, inv) : n; }

// mpeg1video.js:23 / :26 (same text in both passes)
__device__ __forceinline__ void butterfly8(const int (&X)[8], int (&o)[8])
{
    int b1 = X[4];
    int b3 = X[2] + X[6];
    int b4 = X[5] - X[3];
    int tmp1 = X[1] + X[7];
    int tmp2 = X[3] + X[5];
    int b6 = X[1] - X[7];
    int b7 = tmp1 + tmp2;
    int m0 = X[0];
    const int c128 = 128;
    int x4 = div256(mad24k(b6, 473, mad24k(b4, -196, c128))) - b7;
    int x0 = x4 - div256(mad24k(tmp1 - tmp2, 362, c128));
    int x1 = m0 - b1;
    int x2 = div256(mad24k(X[2] - X[6], 362, c128)) - b3;
    int x3 = m0 + b1;
    int y3 = x1 + x2;
    int y4 = x3 + b3;
    int y5 = x1 - x2;
    int y6 = x3 - b3;
    int y7 = -x0 - div256(mad24k(b4, 473, mad24k(b6, 196, c128)));
    o[0] = b7 + y4;
    o[1] = x4 + y3;
    o[2] = y5 - x0;
    o[3] = y6 - y7;
    o[4] = y6 + y7;
    o[5] = x0 + y5;
    o[6] = y3 - x4;
    o[7] = y4 - b7;
}

// The same butterfly with X[2..7] == 0 / X[4..7] == 0 substituted (every dropped term is an exact
// zero: 0*k + 128 divides to 0).  Chosen wave-uniformly when the higher inputs of all 64 lanes
// are zero, which is the common case for quantised video.
__device__ __forceinline__ void butterfly8_lo2(int X0, int X1, int (&o)[8])
{
    const int c128 = 128;
    int x4 = div256(mad24k(X1, 473, c128)) - X1;
    int x0 = x4 - div256(mad24k(X1, 362, c128));
    int y7 = -x0 - div256(mad24k(X1, 196, c128));
    o[0] = X1 + X0;
    o[1] = x4 + X0;
    o[2] = X0 - x0;
    o[3] = X0 - y7;
    o[4] = X0 + y7;
    o[5] = x0 + X0;
    o[6] = X0 - x4;
    o[7] = X0 - X1;
}
__device__ __forceinline__ void butterfly8_lo4(int X0, int X1, int X2, int X3, int (&o)[8])
{
    const int c128 = 128;
    int b7 = X1 + X3;
    int x4 = div256(mad24k(X1, 473, mad24k(X3, 196, c128))) - b7;
    int x0 = x4 - div256(mad24k(X1 - X3, 362, c128));
    int x2 = div256(mad24k(X2, 362, c128)) - X2;
    int y3 = X0 + x2;
    int y4 = X0 + X2;
    int y5 = X0 - x2;
    int y6 = X0 - X2;
    int y7 = -x0 - div256(mad24k(X3, -473, mad24k(X1, 196, c128)));
    o[0] = b7 + y4;
    o[1] = x4 + y3;
    o[2] = y5 - x0;
    o[3] = y6 - y7;
    o[4] = y6 + y7;
    o[5] = x0 + y5;
    o[6] = y3 - x4;
    o[7] = y4 - b7;
}
// dispatch on the number of leading inputs that can be non-zero (wave-uniform)
__device__ __forceinline__ void butterfly8_n(const int (&X)[8], int n_live, int (&o)[8])
{
    if (n_live <= 2) butterfly8_lo2(X[0], X[1], o);
    else if (n_live <= 4) butterfly8_lo4(X[0], X[1], X[2], X[3], o);
    else butterfly8(X, o);
}

// ---- column pass with the last butterfly stage and the hand-off scaling in packed fp32 -----------
// The eight outputs are four (sum, difference) pairs of the same two integers.  Converted to fp32
// (exact: |.| < 2^24 for every input the dequantiser can produce -- DC <= 32767*256, AC <= 2048*62,
// gains < 2 -- so the sums are exact too) they take one v_pk_add_f32 per pair, and the hand-off
// scale by _y = 0.4 one v_pk_mul_f32 per pair: 8 vector instructions less per group than integer
// adds, eight conversions and eight multiplies.
typedef float v2f __attribute__((ext_vector_type(2)));

struct ColOut { v2f p07, p16, p52, p43; };     // (o0,o7) (o1,o6) (o5,o2) (o4,o3)

__device__ __forceinline__ ColOut col_final(int y4, int b7, int y3, int x4, int y5, int x0, int y6, int y7)
{
    const float fy4 = (float)y4, fb7 = (float)b7, fy3 = (float)y3, fx4 = (float)x4;
    const float fy5 = (float)y5, fx0 = (float)x0, fy6 = (float)y6, fy7 = (float)y7;
    ColOut o;
    o.p07 = v2f{fy4, fy4} + v2f{fb7, -fb7};
    o.p16 = v2f{fy3, fy3} + v2f{fx4, -fx4};
    o.p52 = v2f{fy5, fy5} + v2f{fx0, -fx0};
    o.p43 = v2f{fy6, fy6} + v2f{fy7, -fy7};
    return o;
}

__device__ __forceinline__ ColOut butterfly8_col(const int (&X)[8], int n_live)
{
    const int c128 = 128;
    if (n_live <= 2) {
        const int X0 = X[0], X1 = X[1];
        int x4 = div256(mad24k(X1, 473, c128)) - X1;
        int x0 = x4 - div256(mad24k(X1, 362, c128));
        int y7 = -x0 - div256(mad24k(X1, 196, c128));
        return col_final(X0, X1, X0, x4, X0, x0, X0, y7);
    } else if (n_live <= 4) {
        const int X0 = X[0], X1 = X[1], X2 = X[2], X3 = X[3];
        int b7 = X1 + X3;
        int x4 = div256(mad24k(X1, 473, mad24k(X3, 196, c128))) - b7;
        int x0 = x4 - div256(mad24k(X1 - X3, 362, c128));
        int x2 = div256(mad24k(X2, 362, c128)) - X2;
        int y7 = -x0 - div256(mad24k(X3, -473, mad24k(X1, 196, c128)));
        return col_final(X0 + X2, b7, X0 + x2, x4, X0 - x2, x0, X0 - X2, y7);
    } else {
        int b1 = X[4];
        int b3 = X[2] + X[6];
        int b4 = X[5] - X[3];
        int tmp1 = X[1] + X[7];
        int tmp2 = X[3] + X[5];
        int b6 = X[1] - X[7];
        int b7 = tmp1 + tmp2;
        int m0 = X[0];
        int x4 = div256(mad24k(b6, 473, mad24k(b4, -196, c128))) - b7;
        int x0 = x4 - div256(mad24k(tmp1 - tmp2, 362, c128));
        int x1 = m0 - b1;
        int x2 = div256(mad24k(X[2] - X[6], 362, c128)) - b3;
        int x3 = m0 + b1;
        int y7 = -x0 - div256(mad24k(b4, 473, mad24k(b6, 196, c128)));
        return col_final(x3 + b3, b7, x1 + x2, x4, x1 - x2, x0, x3 - b3, y7);
    }
}

// COL_INT_3 (decoders/shaders/mpeg1video.js:21-22) for one coefficient, branch-free.  qO =
// quantiser_scale * matrix entry, pm = premultiplier, nim = -1 for a non-intra block, 0 for an intra
// block.  X*2 (+ sign(X) for non-intra), * qO, floor(./16), even values step toward zero and 0
// becomes +1 (the shader's oddification), clamp to [-2048, 2047], * pm.  A zero coefficient must
// stay zero (the shader's `continue`):
// sign(X) & 1 replaces the constant 1 of the oddification, so X == 0 gives 0*q -> 0 - 0 -> | 0 -> 0
// and every X != 0 gets exactly the reference's arithmetic.
__device__ __forceinline__ int dequant_any(int X, int qO, int pm, int nim, int lo2048, int hi2047)
{
    const int sg = sign3(X);
    int x2 = (X << 1) + (sg & nim);
    int t = mul24_asm(x2, qO);
    int f = t >> 4;
    int z;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(z) : "v"(f));   // inline constants: no registers for 0 and 1
    f = (f - z) | (sg & 1);
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "s"(lo2048), "v"(hi2047));   // one scalar operand is free
    return __mul24(f, pm);
}

// (int)floorf(x) in one instruction
__device__ __forceinline__ int cvt_floor(float x)
{
    int d;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(d) : "v"(x));
    return d;
}

// _B()/_E() int16 hand-off incl. UNORM8 saturation of the high byte (mpeg1video.js:18)
__device__ __forceinline__ int handoff16(int w)
{
    int r = (int)(short)w;                            // |w| < 65536: w mod 2^16
    if (w >= 65536) r = (w & 255) - 256;              // hi byte saturates at 255
    if (w < -65536) r = w & 255;                      // hi byte saturates at 0
    return r;
}

// ---- motion-compensated prediction of 8 horizontally adjacent samples ------

// exact (a+b+c+d+2)>>2 on 4 packed bytes in three v_lerp_u8 (per byte (p + q + (r & 1)) >> 1: the third operand is a
// per-byte rounding bit) and one xor.  With sa = a+b = 2x+rx, sc = c+d = 2y+ry:
//   (sa + sc + 2) >> 2 = (x + y + 1 + (rx & ry)) >> 1,   and   (a + b + ry) >> 1 = x + (rx & ry)   (ry = LSB of c ^ d)
// so X = lerp(a, b, c ^ d), y = lerp(c, d, 0), result = lerp(X, y, 1).  (Round 1 had three rounded-up averages and a
// five-instruction fix-up: twice the instructions.)
__device__ __forceinline__ uint32_t avg4_u8x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t X = __builtin_amdgcn_lerp(a, b, c ^ d);
    const uint32_t y = __builtin_amdgcn_lerp(c, d, 0u);
    return __builtin_amdgcn_lerp(X, y, 0x01010101u);
}

// reference texel addressing: sample x lives in RGBA texel x>>2, component x&3; the
// TEXEL index clamps to the edge (jsv.js:216-217 + _p() mpeg1video.js:24)
__device__ __forceinline__ uint32_t ref_px_clamped(const LEON_GLOBAL uint8_t* ref, uint32_t row_off, int W, int x)
{
    int t = min(max(x >> 2, 0), (W >> 2) - 1);
    return ldg<uint8_t>(ref, row_off + (uint32_t)(4 * t + (x & 3)));
}

typedef v3u U3 __attribute__((aligned(4)));    // 12 bytes that are only dword aligned

// raw reference bytes of one predictor: two rows of 12 bytes starting at the aligned
// address below the window, plus the byte shift of the window inside them
struct RefRows {
    uint32_t l0, l1, l2;        // row y+ay
    uint32_t m0, m1, m2;        // row y+ay+oddv
    uint32_t s;                 // window starts at byte s (0..3)
    uint32_t oh;                // horizontal half-pel flag
};

// 9 samples px..px+8 of one row, for vectors that leave the picture (rare, divergent)
__device__ __forceinline__ void gather9_slow(const LEON_GLOBAL uint8_t* ref, uint32_t row_off, int W, int px,
                                             uint32_t& a0, uint32_t& a1, uint32_t& a2)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        lo |= ref_px_clamped(ref, row_off, W, px + k) << (8 * k);
        hi |= ref_px_clamped(ref, row_off, W, px + 4 + k) << (8 * k);
    }
    a0 = lo;
    a1 = hi;
    a2 = ref_px_clamped(ref, row_off, W, px + 8);
}

// Issue the loads of one predictor: rows y+ay and y+ay+ov, window columns px..px+8.
// px / oh / ov / in_pic come from the task prologue (shared by both halves).
// The lanes (n, b) and (n+1, b) belong to the same block, hence to the same vector: the lower
// row of lane n IS the upper row of lane n+1.  Only the upper row is fetched by every lane;
// the ninth row of a block is fetched by its n == 7 lanes alone (all other lanes carry the
// out-of-range offset, which costs no cache access), and finish_rows() moves the rest
// between lanes.  That halves the L1 accesses of the reference fetch.
__device__ __forceinline__ int clamp_med3(int v, int hi)      // min(max(v, 0), hi) in one instruction
{
    int d;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(d) : "v"(v), "s"(hi));
    return d;
}

// `use` = this lane's macroblock predicts from this reference at all (B pictures: direction map);
// a lane that does not carries the out-of-range offset: no cache access, zeros come back, and
// recon_task() replaces the unused predictor by the other one.
// (The offset of a lane that must not fetch is kOobBit itself, by a select: ADDING the bit to the offset -- tried in round 4
// to save the select -- is wrong for a lane whose unused vector is garbage: its window column is then anything, and the
// sum wraps back into the 2 GiB the resource covers: a real load far outside the allocation.)
__device__ __forceinline__ RefRows fetch_rows(const LEON_GLOBAL uint8_t* ref, int W, int H, int y,
                                              int px, int ay, int oh, int ov, bool in_pic, bool last_row, bool use)
{
    const int yy = y + ay;
    const uint32_t r0 = (uint32_t)__mul24(clamp_med3(yy, H - 1), W);
    // the row below: one line further unless the clamp holds both rows on the same edge line
    const uint32_t r1 = r0 + ((uint32_t)yy < (uint32_t)(H - 1) ? (uint32_t)W : 0u);
    RefRows R;
    R.oh = (uint32_t)oh;
    R.m0 = R.m1 = R.m2 = 0;
    if (in_pic) {
        R.s = (uint32_t)px & 3u;
        uint32_t xo = (uint32_t)px & ~3u;
        // buffer loads: wave-uniform descriptor in SGPRs + 32-bit offset, no 64-bit address math
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc((const void*)ref);
        const v3u a = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)(use && !(LEON_ABL & 1) ? r0 + xo : kOobBit), 0, 0);
        const v3u c = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)(use && last_row && ov && !(LEON_ABL & 1) ? r1 + xo : kOobBit), 0, 0);
        R.l0 = a.x; R.l1 = a.y; R.l2 = a.z;
        R.m0 = c.x; R.m1 = c.y; R.m2 = c.z;
    } else {                                         // vector leaves the picture (rare)
        R.s = 0;
        gather9_slow(ref, r0, W, px, R.l0, R.l1, R.l2);
        if (last_row && ov) gather9_slow(ref, r1, W, px, R.m0, R.m1, R.m2);
    }
    return R;
}

// second half of fetch_rows: the lower row comes from the lane 8 above (same block, next sample
// row) -- or from the lane itself when the vector has no vertical half-pel part
// `below` (luma, upper half): the half underneath belongs to the same macroblocks, its first row IS the ninth row of this one -- same
// vector, same clamped row address (fetch_rows: r1 of row y equals r0 of row y + 1 at every edge) -- so the upper half fetches no ninth
// row of its own: its last lanes take it from the lower half's first lanes ((lane + 8) & 63 is that lane).
__device__ __forceinline__ void finish_rows(RefRows& R, int ov, bool last_row, int lane, const RefRows* below = nullptr)
{
    const int src = (ov ? ((lane + 8) & 63) : lane) << 2;
    const uint32_t n0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)R.l0);
    const uint32_t n1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)R.l1);
    const uint32_t n2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)R.l2);
    if (below) {
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)below->l0);
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)below->l1);
        const uint32_t b2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)below->l2);
        const bool wrap = last_row && ov;
        R.m0 = wrap ? b0 : n0; R.m1 = wrap ? b1 : n1; R.m2 = wrap ? b2 : n2;
        return;
    }
    if (!(last_row && ov)) { R.m0 = n0; R.m1 = n1; R.m2 = n2; }
}

// (a+b+c+d+2)>>2 / (a+b+1)>>1 / a, selected by the half-pel flags through operand
// duplication: a==b when !oh (shift 0), rows equal when !ov (same row fetched twice)
__device__ __forceinline__ v2u predict8(const RefRows& R)
{
    uint32_t a0 = __builtin_amdgcn_alignbyte(R.l1, R.l0, R.s);
    uint32_t a1 = __builtin_amdgcn_alignbyte(R.l2, R.l1, R.s);
    uint32_t a2 = __builtin_amdgcn_alignbyte(0u, R.l2, R.s);
    uint32_t c0 = __builtin_amdgcn_alignbyte(R.m1, R.m0, R.s);
    uint32_t c1 = __builtin_amdgcn_alignbyte(R.m2, R.m1, R.s);
    uint32_t c2 = __builtin_amdgcn_alignbyte(0u, R.m2, R.s);
    uint32_t b0 = __builtin_amdgcn_alignbyte(a1, a0, R.oh);
    uint32_t b1 = __builtin_amdgcn_alignbyte(a2, a1, R.oh);
    uint32_t d0 = __builtin_amdgcn_alignbyte(c1, c0, R.oh);
    uint32_t d1 = __builtin_amdgcn_alignbyte(c2, c1, R.oh);
    v2u p;
    p.x = avg4_u8x4(a0, b0, c0, d0);
    p.y = avg4_u8x4(a1, b1, c1, d1);
    return p;
}

// byte M of `pred` moved to bits 8..15 (so that it adds as pred*256), one v_perm_b32
template <int M>
__device__ __forceinline__ uint32_t pred_x256(uint32_t pred)
{
    // selector bytes: 0x0c = constant 0x00; 0..3 pick bytes of the second operand.  The selector
    // rides in a scalar register (a VOP3 instruction may read one): the builtin makes the compiler
    // re-materialise it in a vector register in front of every group of uses.
    uint32_t d;
    asm("v_perm_b32 %0, 0, %1, %2" : "=v"(d) : "v"(pred), "s"(0x0c0c000cu | ((uint32_t)M << 8)));
    return d;
}

// ---- one task = two block groups that share their macroblocks ------------------------
//   luma  : the upper and lower 64x8 halves of four macroblocks (block rows 2*Rt, 2*Rt+1)
//   chroma: the Cb and the Cr group at block row Rt
// Everything that depends only on the macroblock -- maps, vectors and their half-pel
// decomposition, quantiser tables, the in-picture test -- is computed once per task.

// Fused display conversion (DISPLAY): the picture leaves the kernel as RGBA8 as well -- the CPU twin of
// the reference's conversion, bit for bit what k_rgba_twin4 below computes in fp64 (here from exact fixed-point
// tables, see rgba_px) -- so the planes of a picture nobody predicts from (B pictures: 8 of 12 in IBBP) are never written and never
// read back, and the others are not read back.  A task then covers 8 macroblocks completely: first the
// chroma part (CHROMA = true: the Cb and Cr groups, whose samples are parked in an LDS stash), then the
// two luma parts (CHROMA = false, 4 macroblocks each), whose lanes hold 8 horizontally adjacent Y samples
// and find their 4 Cb and 4 Cr samples in the stash.
struct Display {
    char* stash;                 // LDS: [Cb | Cr][8 rows][64 bytes]
    int side;                    // luma parts: 0 / 1 = left / right four macroblocks of the chroma group
    char* apark;                 // yuva: the A samples of the four macroblocks, [half][8 rows][64 bytes]
    const char* lut;             // LDS copy of Tables::rgba_lut (the workgroup's)
};
// AMODE of recon_task in a yuva display task: the A part runs before the Y part of the same four macroblocks
// and parks its samples (1); the Y part's conversion takes its alpha bytes from there (2); 0 otherwise.

// byte i of a packed dword, times 2^sh: one v_lshlrev_b32_sdwa (table addresses from packed samples)
template <int I>
__device__ __forceinline__ uint32_t byte_shl(uint32_t v, uint32_t sh)
{
    uint32_t d;
    if constexpr (I == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(sh), "v"(v));
    else if constexpr (I == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(sh), "v"(v));
    else if constexpr (I == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(sh), "v"(v));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(sh), "v"(v));
    return d;
}

// The reference's YCbCrToRGBA (player/easybits.player.js:2692-2782) on fixed-point tables: every term of
//   R = store(r(Cr) + ys(Y)), G = store((g1(Cr) - g2(Cb)) + ys(Y)), B = store(b(Cb) + ys(Y))
// (store = Uint8ClampedArray: clamp, round half to even) comes out of an LDS table with 21 fractional bits,
// the sum is shifted, clamped and packed by v_ashr_pk_u8_i32 -- and equals the fp64 arithmetic for all 2^24
// inputs (tools/make_rgba_lut.py explains why and checks it; tests/test_rgba_lut.py, and the exhaustive GPU
// test in tests/test_fused_display_gpu.py).  7 vector instructions per pixel instead of 20 in fp64.
struct ChromaTerms { int r, g, b; };
template <int I>
__device__ __forceinline__ ChromaTerms chroma_terms(const char* lut, uint32_t cb2, uint32_t cr2, uint32_t three)
{
    const v2u rc = *reinterpret_cast<const v2u*>(lut + 1024 + byte_shl<I>(cr2, three));     // {r, g1}[Cr]
    const v2u bc = *reinterpret_cast<const v2u*>(lut + 3072 + byte_shl<I>(cb2, three));     // {b, -g2}[Cb]
    return ChromaTerms{(int)rc.x, (int)(rc.y + bc.y), (int)bc.x};
}
template <int I>
__device__ __forceinline__ uint32_t rgba_px(const char* lut, uint32_t y4, const ChromaTerms& c, int a_fixed, uint32_t two)
{
    const int ys = *reinterpret_cast<const int*>(lut + byte_shl<I>(y4, two));
    // {sat_u8(a >> 21), sat_u8(b >> 21)} go to the low half of the destination, and -- op_sel[3] -- of the second
    // instruction to the high half, the other half left alone: R G B A packed by the two shifts themselves (no byte
    // permute behind them).  Through asm: the builtin knows neither the half-selecting form nor that the other half stays.
    uint32_t px;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(px) : "v"(ys + c.r), "v"(ys + c.g), "n"(kLutShift));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(px) : "v"(ys + c.b), "v"(a_fixed), "n"(kLutShift));
    return px;
}
// 4 horizontally adjacent pixels of one row: Y samples y4, the two chroma pairs' terms, A samples a4
template <bool ALPHA>
__device__ __forceinline__ v4u rgba_row4(const char* lut, uint32_t y4, const ChromaTerms& c0, const ChromaTerms& c1, uint32_t a4,
                                         uint32_t two, uint32_t k21)
{
    const int opaque = 255 << kLutShift;
    int a[4] = {opaque, opaque, opaque, opaque};
    if constexpr (ALPHA) {
        a[0] = (int)byte_shl<0>(a4, k21); a[1] = (int)byte_shl<1>(a4, k21);
        a[2] = (int)byte_shl<2>(a4, k21); a[3] = (int)byte_shl<3>(a4, k21);
    }
    return v4u{rgba_px<0>(lut, y4, c0, a[0], two), rgba_px<1>(lut, y4, c0, a[1], two),
               rgba_px<2>(lut, y4, c1, a[2], two), rgba_px<3>(lut, y4, c1, a[3], two)};
}

// ---- stage 5 (fused display conversion): RGBA of one half (64 x 8 samples) of a luma part ---------------
// A lane of recon_task holds samples 8b .. 8b+7 of row n; converted as they lie, a store instruction would
// write 16-byte pieces 32 bytes apart (half-filled lines) and the two rows that share a chroma row would
// sit in different lanes.  So the rows went through LDS (`ypark`: [8 rows][64 bytes], written at the end of
// the half and ordered by the fences there) and lane (n, b) converts the pixel quad j = 2b + (n & 1) of the
// row pair p = n >> 1: pixels 4j .. 4j+3 of rows 2p and 2p+1, whose chroma terms it looks up once; a store
// instruction writes 256 contiguous bytes of each of four rows = full 128-byte lines.
template <int AMODE>
__device__ __forceinline__ void display_half(const PicDesc& pd, const Geom& G, const Display& dsp, const char* ypark,
                                             int half, int Rt, int g, int hi3, int lo3)
{
    const int pr = hi3 >> 1, jq = 2 * lo3 + (hi3 & 1);
    const int xa = 64 * g + 4 * jq;                            // first pixel of the lane's quad
    const __amdgpu_buffer_rsrc_t rrs = buf_rsrc(pd.rgba);
    uint32_t two = 2u, three = 3u, k21 = (uint32_t)kLutShift;  // SDWA shift counts live in registers
    asm("" : "+v"(two), "+v"(three));
    if constexpr (AMODE == 2) asm("" : "+v"(k21));
    const int yrow = 8 * (2 * Rt + half) + 2 * pr;
    // chroma: row y>>1 = stash row 4*half + p; the quad's two chroma samples are bytes 32*side + 2j, + 1
    const char* yp = ypark + pr * 128 + jq * 4;
    const char* sp = dsp.stash + (4 * half + pr) * 64 + 32 * dsp.side + 2 * jq;
    const uint32_t ya = *reinterpret_cast<const uint32_t*>(yp), yb = *reinterpret_cast<const uint32_t*>(yp + 64);
    const uint32_t cb2 = *reinterpret_cast<const uint16_t*>(sp), cr2 = *reinterpret_cast<const uint16_t*>(sp + 512);
    uint32_t aa = 0u, ab = 0u;                             // yuva: the pixels' A samples, parked by the A part
    if constexpr (AMODE == 2) {
        const char* ap = dsp.apark + half * 512 + pr * 128 + jq * 4;
        aa = *reinterpret_cast<const uint32_t*>(ap);
        ab = *reinterpret_cast<const uint32_t*>(ap + 64);
    }
    const ChromaTerms c0 = chroma_terms<0>(dsp.lut, cb2, cr2, three), c1 = chroma_terms<1>(dsp.lut, cb2, cr2, three);
    // the frame is the top-left crop of the coded picture; its width is a multiple of 8 (host check)
    const uint32_t row_off = __umul24((uint32_t)yrow, (uint32_t)G.fw) + (uint32_t)xa;     // both < 4096
    const bool in_a = yrow < G.fh && xa < G.fw, in_b = yrow + 1 < G.fh && xa < G.fw;
    const v4u pa = rgba_row4<AMODE == 2>(dsp.lut, ya, c0, c1, aa, two, k21);
    if (!(LEON_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(pa, rrs, (int)((row_off * 4u) | (in_a ? 0u : kOobBit)), 0, LEON_RGBA_AUX);
    const v4u pb = rgba_row4<AMODE == 2>(dsp.lut, yb, c0, c1, ab, two, k21);
    if (!(LEON_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(pb, rrs, (int)(((row_off + (uint32_t)G.fw) * 4u) | (in_b ? 0u : kOobBit)), 0, LEON_RGBA_AUX);
    if (LEON_ABL & 2) asm volatile("" :: "v"(pa.x ^ pa.y ^ pa.z ^ pa.w ^ pb.x ^ pb.y ^ pb.z ^ pb.w));      // keep the conversion alive
}

// ---- stage 2 as functions: the liveness scan of a tile and the column pass over a list of live columns ------------------------
// A column id: tile << 7 | half << 6 | c << 3 | b.  scan_tile: lane (c = hi3, b = lo3) looks at its own column of each half of
// `tile`; the live lanes queue up behind the n_before ids that are in the list already.  Returns the new length.
__device__ __forceinline__ uint32_t scan_tile(const char* tile, char* ids, int lane, uint32_t tag, uint32_t n_before, uint64_t (&live)[2])
{
    const int hi3 = lane >> 3, lo3 = lane & 7;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const char* cp = tile + h * kLdsHalf + lo3 * 16 + hi3 * 2;
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= *reinterpret_cast<const uint16_t*>(cp + i * 128);
        live[h] = lanes_nonzero((int)o);
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(live[h] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live[h], n_before));
        if (o != 0u) *reinterpret_cast<uint8_t*>(ids + pos) = (uint8_t)(tag + (uint32_t)lane + 64u * h);
        n_before += (uint32_t)__builtin_popcountll(live[h]);
    }
    return n_before;
}

// The column pass (COL_INT_3 / COL_INT_5, decoders/shaders/mpeg1video.js:19-24) over the n_cols live columns listed in `ids`:
// lane k takes the k-th; the results replace the coefficients in place.  PAIR: ids may name the second tile (tile0 + tile_step),
// whose blocks' quantiser scale and intra flag sit in qia1 (lane b: block b's) as the first tile's in qia0.
// TLDS: the tables are the wave's LDS copy at `qtab`; else they are in `qreg`: the picture's QTables, 256 bytes, dword i in lane i
// -- a lane fetches the two dwords of its column's matrix row and of its premultiplier row with four ds_bpermute.  (Until the end of
// round 4 the two-tile layout, which has no room for the tables in LDS, read them from memory inside this loop: two loads and an
// `s_waitcnt vmcnt(0)` per round -- and loads return in order: the chroma part's wait for its tables was a wait for the REFERENCE
// rows it had requested just before, whose flight the column pass was meant to cover.)
template <bool PAIR, bool TLDS = true>
__device__ __forceinline__ void column_pass(char* tile0, uint32_t tile_step, const char* ids, const char* qtab, uint32_t n_cols, int qia0, int qia1, int lane,
                                            uint32_t qreg = 0u)
{
    // the clamp bounds live in registers (v_med3 takes no literals on gfx950, and the compiler would
    // otherwise re-materialise them in front of every use): one scalar, one vector -- a VOP3
    // instruction may read one scalar register
    int lo2048 = -2048, hi2047 = 2047;
    asm("" : "+s"(lo2048), "+v"(hi2047));
#pragma unroll 1
    for (uint32_t base = 0; base < n_cols; base += 64u) {             // wave-uniform: once, twice for busy I pictures
        const uint32_t k = base + (uint32_t)lane;
        const bool act = k < n_cols;
        uint32_t id = *reinterpret_cast<const uint8_t*>(ids + k);
        id = act ? id : 0u;                               // idle lanes redo column 0 and write nothing
        char* colp = tile0 + ((id & 0x47u) << 4) + ((id >> 2) & 14u);
        if constexpr (PAIR) colp += (id >> 7) * tile_step;
        int X[8];
#pragma unroll
        for (int i = 0; i < 8; i++) X[i] = *reinterpret_cast<const short*>(colp + i * 128);
        int bq = __builtin_amdgcn_ds_bpermute((int)((id & 7u) << 2), qia0);      // lane b holds block b's macroblock
        if constexpr (PAIR) {
            const int bq1 = __builtin_amdgcn_ds_bpermute((int)((id & 7u) << 2), qia1);
            bq = (id & 128u) ? bq1 : bq;
        }
        const bool bia = bq >= 256;
        const uint32_t qv = (uint32_t)bq & 31u;
        v2u msel, pm8;                                                               // column c of the tables
        if constexpr (TLDS) {
            const char* const qt = qtab + (id & 56u);
            msel = *reinterpret_cast<const v2u*>(qt + (bia ? 0 : 64));
            pm8 = *reinterpret_cast<const v2u*>(qt + 128);
        } else {
            const int qo = (int)((id & 56u) + (bia ? 0u : 64u)), po = (int)((id & 56u) + 128u);      // byte offsets = 4 x the lane that holds the dword
            msel.x = (uint32_t)__builtin_amdgcn_ds_bpermute(qo, (int)qreg);
            msel.y = (uint32_t)__builtin_amdgcn_ds_bpermute(qo + 4, (int)qreg);
            pm8.x = (uint32_t)__builtin_amdgcn_ds_bpermute(po, (int)qreg);
            pm8.y = (uint32_t)__builtin_amdgcn_ds_bpermute(po + 4, (int)qreg);
        }
        int nim = bia ? 0 : -1;
        asm("" : "+v"(nim));                              // keep it a mask (v_and), not a select
        const bool dc_lane = bia && (id & 56u) == 0u;
        uint64_t nz[8];
#pragma unroll
        for (int i = 0; i < 8; i++) nz[i] = lanes_nonzero(X[i]);
        const int dc = X[0];
        int rows_live = 1;                                // wave-uniform: 1 + highest row with a non-zero
#pragma unroll
        for (int i = 0; i < 8; i++) {
            // a row whose coefficients are zero in every column of the pass costs its compare and a scalar
            // branch; in a live row every lane runs the branch-free form (zeros stay zero)
            if (nz[i] != 0) {
                rows_live = i + 1;
                int P = (int)(((i < 4 ? pm8.x : pm8.y) >> (8 * (i & 3))) & 255u);
                const int qO = (int)__umul24(qv, ((i < 4 ? msel.x : msel.y) >> (8 * (i & 3))) & 255u);   // quantiser_scale * Q[i][c] < 2^13
                X[i] = dequant_any(X[i], qO, P, nim, lo2048, hi2047);
            }
        }
        if (dc_lane) X[0] = dc * 256;                     // COL_4 / COL_INT_31
        const ColOut co = butterfly8_col(X, rows_live);
        // floor( float(v) * _y ): the int16 the reference hands from pass 1 to pass 2
        const v2f k04 = {0.4f, 0.4f};
        const v2f s07 = co.p07 * k04, s16 = co.p16 * k04, s52 = co.p52 * k04, s43 = co.p43 * k04;
        // floor and conversion in one instruction (v_cvt_flr_i32_f32)
        int wi[8] = {cvt_floor(s07.x), cvt_floor(s16.x), cvt_floor(s52.y), cvt_floor(s43.y),
                     cvt_floor(s43.x), cvt_floor(s52.x), cvt_floor(s16.y), cvt_floor(s07.y)};
        // |s| < 32768 for all eight: every floor(s) is an int16 as it stands
        const float mx = fmaxf(fmaxf(fmaxf(fabsf(s07.x), fabsf(s16.x)), fmaxf(fabsf(s52.y), fabsf(s43.y))),
                               fmaxf(fmaxf(fabsf(s43.x), fabsf(s52.x)), fmaxf(fabsf(s16.y), fabsf(s07.y))));
        if (mx >= 32768.0f) {                             // outside any real stream: int16 wrap / saturation
#pragma unroll
            for (int n = 0; n < 8; n++) wi[n] = handoff16(wi[n]);
        }
        if (act) {
#pragma unroll
            for (int n = 0; n < 8; n++) *reinterpret_cast<short*>(colp + n * 128) = (short)wi[n];
        }
    }
}

// `alpha` (wave-uniform, luma-shaped tasks only): the task reconstructs the A plane of a yuva picture --
// the same code path as luma with its own coefficient plane, the plane behind Cr in every slot, and the
// alpha groups of the sparse lists.
// CARRY (display tasks): the eight macroblocks of a task are looked up in the maps ONCE, by its chroma part (1: lane
// (., m) loads macroblock m's quantiser scale, flags and vectors as always and leaves them in `carry`); the luma and
// alpha parts (2) take theirs from the lane of their macroblock by ds_bpermute -- no loads, no second wait for memory
// in front of their reference fetches.  0: a task on its own loads what it needs.
struct MbCarry { uint32_t flags; };       // q | intra << 8 | repadd >= 128 << 9 | direction << 10 (the vectors: LDS, kOffCarry)

// BACK (dense display tasks whose two luma parts share their front, recon_luma_pair): the part's coefficients are in `tile` and have
// been through the column pass already; `live_in` says which of its columns were live
template <int TYPE, bool CHROMA, bool SPARSE, bool DISPLAY, int AMODE = 0, int CARRY = 0, bool BACK = false, int LAYOUT = 0>
__device__ __forceinline__ void recon_task(const PicDesc& pd, const Geom& G, int Rt, int g, char* lds, int lane, Display dsp, MbCarry& carry, bool alpha = false,
                                           char* tile_in = nullptr, uint64_t live_in0 = 0, uint64_t live_in1 = 0, uint32_t qreg = 0u)
{
    char* const tile = BACK ? tile_in : lds;
    const int W = CHROMA ? G.cw >> 1 : G.cw;
    const int H = CHROMA ? G.ch >> 1 : G.ch;
    const int bw = W >> 3;
    const uint32_t ysz = (uint32_t)G.cw * (uint32_t)G.ch;
    const uint32_t a_off = !CHROMA && alpha ? ysz + (ysz >> 1) : 0u;       // byte offset of the A plane in a slot
    const int hi3 = lane >> 3, lo3 = lane & 7;

    // ---- shared prologue ----------------------------------------------------------------
    const int Qld = g * 8 + lo3;                      // stage-1 role: lane (r = hi3, b = lo3)
    const bool ld_ok = Qld < bw;
    // afterwards: lane (n = hi3, b = lo3) in the row pass and behind it: the 8 lanes of an aligned
    // group hold the 8 blocks of ONE sample row, so the reference fetches and the stores of a group are
    // contiguous along a picture row and the texture addresser merges them into 64-byte accesses (with
    // rows across adjacent lanes every lane was its own L1 access, and the vector L1 -- one access per
    // clock -- was the limiter)
    const int b = lo3;
    const int Qb = g * 8 + b;
    const bool valid = Qb < bw;
    const int Qs = valid ? Qb : bw - 1;
    // everything that does not depend on the macroblock maps is requested first: the coefficient rows of
    // BOTH halves (dense boundary: straight into the tile) or the first entries of both runs (sparse)
    const int R0 = CHROMA ? Rt : 2 * Rt;
    const uint32_t coef_voff = (2u * ((uint32_t)__mul24(8 * R0 + hi3, W) + (uint32_t)(8 * Qld))) | (ld_ok ? 0u : kOobBit);
    const uint32_t half_step = CHROMA ? 0u : 8u * (uint32_t)W;   // luma: next block row; chroma: next plane
    // sparse boundary: the two groups of the task are two runs of entries[]; the first 64
    // entries of each are requested here (one dword per lane), longer runs loop in stage 1
    uint32_t ent_first[2] = {0u, 0u}, ent_start[2] = {0u, 0u}, ent_count[2] = {0u, 0u};
    // a resource of exactly n_entries dwords: indices past the list read 0, whatever grp_off says
    const __amdgpu_buffer_rsrc_t ent_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(SPARSE ? pd.entries : nullptr), 0, SPARSE ? (int)(pd.n_entries * 4u) : 0, 0x00020000);
    if constexpr (SPARSE) {
        const uint32_t nY = 2u * (uint32_t)G.tasksY, nC = (uint32_t)G.tasksC;
        const uint32_t gid0 = CHROMA ? nY + (uint32_t)(Rt * G.gC + g) : (uint32_t)(2 * Rt * G.gY + g) + (alpha ? nY + 2u * nC : 0u);
        const uint32_t gid1 = CHROMA ? gid0 + nC : gid0 + (uint32_t)G.gY;
        const LEON_GLOBAL uint32_t* go = gptr(pd.grp_off);    // wave-uniform index; global, not flat: a flat load counts on the LDS counter too
        const uint32_t gid[2] = {gid0, gid1};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t s0 = go[gid[h]], e0 = go[gid[h] + 1];
            ent_start[h] = s0;
            ent_count[h] = e0 > s0 ? min(e0 - s0, 512u) : 0u;      // a group holds at most 8*64 coefficients
            ent_first[h] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                ent_rs, (int)(((s0 + (uint32_t)lane) * 4u) | ((uint32_t)lane < ent_count[h] ? 0u : kOobBit)), 0, kAuxStreamOnce);
        }
    } else if constexpr (!BACK) {
        // the previous task of this wave may still be reading the tile
        wait_lds_all();
        __builtin_amdgcn_wave_barrier();
        coef_rows_to_lds(pd.coef[CHROMA ? 1 : (alpha ? 3 : 0)], lds, coef_voff, 0u);
        coef_rows_to_lds(pd.coef[CHROMA ? 2 : (alpha ? 3 : 0)], lds + kLdsHalf, coef_voff, 2u * half_step);
    }
    const uint32_t mb = (uint32_t)(CHROMA ? Rt * G.mbw + Qs : Rt * G.mbw + (Qs >> 1));
    uint32_t mf = 0, mk = 0, flags;
    if constexpr (CARRY == 2) {
        // this lane's macroblock is number 4 * side + (block >> 1) of the task: its chroma-part lane holds it
        const int src = (4 * dsp.side + (lo3 >> 1)) << 2;
        flags = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)carry.flags);
        // the vectors wait in LDS (the 64 bytes behind the staged tables: kOffCarry) -- carried in registers like the
        // flags, the three words went to scratch memory in the B kernel (12 bytes per lane: +6 % HBM traffic per launch)
        if (TYPE != 1) mf = *reinterpret_cast<const uint32_t*>(lds + Lay<LAYOUT>::carry + src);
        if (TYPE == 3) mk = *reinterpret_cast<const uint32_t*>(lds + Lay<LAYOUT>::carry + 32 + src);
    } else {
        flags = (uint32_t)(ldg<uint8_t>(gptr(pd.qscale), mb) & 31) | (ldg<uint8_t>(gptr(pd.intra), mb) != 0 ? 256u : 0u);   // I pictures honour the intra map too (COL_3)
        if (TYPE != 1) {
            flags |= ldg<uint8_t>(gptr(pd.repadd), mb) >= 128 ? 512u : 0u;    // .r > 0.5
            mf = ldg<uint32_t>(gptr(pd.mv_fwd), mb * 4);
        }
        if (TYPE == 3) {
            mk = ldg<uint32_t>(gptr(pd.mv_bwd), mb * 4);
            flags |= (uint32_t)(ldg<uint8_t>(gptr(pd.mb_dir), mb) & 3) << 10;
        }
        if constexpr (CARRY == 1) {       // lane i < 8 holds macroblock i of the task (and so does every lane i + 8 k)
            carry.flags = flags;
            if (TYPE != 1) *reinterpret_cast<uint32_t*>(lds + Lay<LAYOUT>::carry + ((lane & 7) << 2)) = mf;
            if (TYPE == 3) *reinterpret_cast<uint32_t*>(lds + Lay<LAYOUT>::carry + 32 + ((lane & 7) << 2)) = mk;
        }
    }
    const int q = (int)(flags & 31u);
    const bool ia = (flags & 256u) != 0u;
    const int x0 = 8 * Qs;
    bool nopred = false;
    // per reference: window column, half-pel flags, vertical offset, base selection
    int pxA = 0, ayA = 0, ohA = 0, ovA = 0, pxB = 0, ayB = 0, ohB = 0, ovB = 0;
    bool inA = true, inB = true, usef = true, useb = true;
    if (TYPE != 1) {
        nopred = (flags & 512u) != 0u;
        if (TYPE == 3) {
            const int dir = (int)(flags >> 10) & 3;
            usef = (dir & 1) != 0;
            useb = (dir & 2) != 0;
            nopred = nopred || dir == 0;
            // predictor A always reads the forward reference, B the backward one (scalar bases,
            // 32-bit offsets); an unused direction fetches nothing (out-of-range offset in
            // fetch_rows; its vector is zeroed so that the in-picture fast path is taken) and is
            // replaced by the other predictor afterwards: (p + p + 1) >> 1 == p
            if (!usef) mf = 0;                                // both components at once
            if (!useb) mk = 0;
        }
        const int fh = (int)(short)(mf & 0xffff), fv = (int)mf >> 16;
        const int bh = (int)(short)(mk & 0xffff), bv = (int)mk >> 16;
        {   // chroma: vector truncated toward zero first (mv_coef 0.5), then floor / parity
            int h = CHROMA ? fh / 2 : fh, v = CHROMA ? fv / 2 : fv;
            pxA = x0 + (h >> 1); ohA = h & 1; ayA = v >> 1; ovA = v & 1;
            inA = (uint32_t)pxA < (uint32_t)(W - 7 - ohA);    // 0 <= px && px + 7 + oh <= W - 1 (W >= 8)
        }
        if (TYPE == 3) {
            int h = CHROMA ? bh / 2 : bh, v = CHROMA ? bv / 2 : bv;
            pxB = x0 + (h >> 1); ohB = h & 1; ayB = v >> 1; ovB = v & 1;
            inB = (uint32_t)pxB < (uint32_t)(W - 7 - ohB);
        }
    }
    // lanes that take nothing from a reference: the direction map says so (B), or the macroblock is
    // not predicted at all (RepAdd / no direction).  They fetch nothing, and never take the slow path.
    const bool useA = usef && !nopred, useB = useb && !nopred;
    inA = inA || !useA;
    inB = inB || !useB;
    // quantiser scale and intra flag of this lane's block, for the lanes of the column pass to fetch (ds_bpermute)
    const int qia = q | (ia ? 256 : 0);

    // per-lane offsets of the task's first half; the second half differs by a scalar
    const uint32_t out_voff = ((uint32_t)__mul24(8 * R0 + hi3, W) + (uint32_t)x0) | (valid ? 0u : kOobBit);
    RefRows rfh[2], rbh[2];
    // B pictures: a wave whose macroblocks all predict from one side only (the leading pictures of
    // a closed GOP, runs of forward- or backward-only macroblocks) neither fetches nor interpolates
    // the other reference.  Wave-uniform, so it costs two scalar tests.
    const bool any_f = TYPE != 3 || __builtin_amdgcn_ballot_w64(useA) != 0;
    const bool any_b = TYPE == 3 && __builtin_amdgcn_ballot_w64(useB) != 0;
    // the coefficient rows (requested before the maps the reference fetches wait for anyway) have landed
    if constexpr (!SPARSE && !BACK) wait_vmem_all();
    if (TYPE != 1) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int Rh = CHROMA ? Rt : 2 * Rt + h;
            const uint32_t po = CHROMA ? (h == 0 ? ysz : ysz + (ysz >> 2)) : a_off;
            // (luma, upper half: no ninth row of its own -- finish_rows takes it from the lower half)
            const bool ninth = hi3 == 7 && (CHROMA || h == 1 || !LEON_NINTH_FROM_BELOW);
            if (any_f) rfh[h] = fetch_rows(gptr(pd.ref_fwd) + po, W, H, 8 * Rh + hi3, pxA, ayA, ohA, ovA, inA, ninth, useA);
            if (TYPE == 3 && any_b) rbh[h] = fetch_rows(gptr(pd.ref_bwd) + po, W, H, 8 * Rh + hi3, pxB, ayB, ohB, ovB, inB, ninth, useB);
        }
    }

    // ---- stage 1: the tile [half][r][b][c] ---------------------------------------------------------
    // dense: the rows are there (LDS-direct loads, waited for above); sparse: clear the tile and scatter both runs
    if constexpr (SPARSE) {
        *reinterpret_cast<v4i*>(lds + lane * 16) = v4i{0, 0, 0, 0};
        *reinterpret_cast<v4i*>(lds + kLdsHalf + lane * 16) = v4i{0, 0, 0, 0};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // scatter the group's entries into the cleared tile; the offset is masked to the tile
            uint32_t e = ent_first[h];
            if (e != 0u) *reinterpret_cast<short*>(lds + h * kLdsHalf + ((e >> 16) & 1022u)) = (short)e;
#pragma unroll 1
            for (uint32_t k = 64u; k < ent_count[h]; k += 64u) {     // wave-uniform, rare
                const uint32_t idx = k + (uint32_t)lane;
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                    ent_rs, (int)(((ent_start[h] + idx) * 4u) | (idx < ent_count[h] ? 0u : kOobBit)), 0, kAuxStreamOnce);
                if (e != 0u) *reinterpret_cast<short*>(lds + h * kLdsHalf + ((e >> 16) & 1022u)) = (short)e;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage 2: column pass over the LIVE columns of both halves ----------------------------------
    // A column (half, block, c) without a coefficient gives eight zeros, and zeros are what it holds already.
    // Quantised video leaves few columns alive (one in ten in P and B pictures, a third in I pictures), so
    // the lanes first find the live ones -- lane (c = hi3, b = lo3) looks at its own column of each half --
    // and the wave then runs the pass on them alone, usually in one go for both halves instead of one go
    // per half with most lanes computing zeros.  The results replace the coefficients in place.
    uint64_t live[2] = {live_in0, live_in1};
    if constexpr (!BACK) {
        const uint32_t n_cols = scan_tile(tile, lds + Lay<LAYOUT>::slots, lane, 0u, 0u, live);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        column_pass<false, Lay<LAYOUT>::tables_in_lds>(tile, 0u, lds + Lay<LAYOUT>::slots, Lay<LAYOUT>::tables_in_lds ? lds + kOffQtab : reinterpret_cast<const char*>(pd.qt),
                                                       n_cols, qia, qia, lane, qreg);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

#pragma unroll
    for (int half = 0; half < 2; half++) {
        const uint32_t plane_off = CHROMA ? (half == 0 ? ysz : ysz + (ysz >> 2)) : a_off;
        RefRows rf = rfh[half], rb = rbh[half];
        int t[8];
        const uint64_t colbits = live[half];
        if (colbits == 0) {
            // no coefficient in any of the 8 blocks: residual 0 = (0 + 128) / 256
#pragma unroll
            for (int m = 0; m < 8; m++) t[m] = 128;
        } else {
            // ---- stage 3: row pass ---------------------------------------------------------
            // lane (n = hi3, b = lo3): row n of block b, eight int16 w; int( w / _y ) == trunc(5w/2) == trunc(w * 2.5f)
            // (exact products, the conversion truncates).  Columns that are dead in every block of the half give
            // zero inputs (wave-uniform): lane (c, b) of the liveness mask -> byte c
            const int cols_live = 8 - (__builtin_clzll(colbits | 1ull) >> 3);
            const v4u wv = *reinterpret_cast<const v4u*>(tile + half * kLdsHalf + hi3 * 128 + lo3 * 16);
            const v2f k25 = {2.5f, 2.5f};
            const v2f a = v2f{(float)(short)(wv.x & 0xffffu), (float)((int)wv.x >> 16)} * k25;
            const int Y0 = (int)a.x + 128, Y1 = (int)a.y;                           // "+128" of (t+128)/256
            if (cols_live <= 2) {
                butterfly8_lo2(Y0, Y1, t);
            } else {
                const v2f bb = v2f{(float)(short)(wv.y & 0xffffu), (float)((int)wv.y >> 16)} * k25;
                if (cols_live <= 4) {
                    butterfly8_lo4(Y0, Y1, (int)bb.x, (int)bb.y, t);
                } else {
                    const v2f cc = v2f{(float)(short)(wv.z & 0xffffu), (float)((int)wv.z >> 16)} * k25;
                    const v2f dd = v2f{(float)(short)(wv.w & 0xffffu), (float)((int)wv.w >> 16)} * k25;
                    const int Y[8] = {Y0, Y1, (int)bb.x, (int)bb.y, (int)cc.x, (int)cc.y, (int)dd.x, (int)dd.y};
                    butterfly8(Y, t);
                }
            }
            // t/256 truncating == arithmetic shift after adding 255 to negative values.  Without a
            // prediction (I pictures) the difference between truncation and floor is invisible:
            // it only exists for negative t, which the final clamp turns into 0 either way.
            if (TYPE != 1) {
#pragma unroll
                for (int m = 0; m < 8; m++) t[m] += (t[m] >> 31) & 255;
            }
        }

        // ---- stage 4: prediction, add, clamp, store ------------------------------------------
        if (TYPE != 1) {
            v2u pred = {0u, 0u};
            if (any_f) {
                finish_rows(rf, ovA, hi3 == 7, lane, !CHROMA && half == 0 && LEON_NINTH_FROM_BELOW ? &rfh[1] : nullptr);
                pred = predict8(rf);
            }
            if (TYPE == 3) {
                v2u pb = {0u, 0u};
                if (any_b) {
                    finish_rows(rb, ovB, hi3 == 7, lane, !CHROMA && half == 0 && LEON_NINTH_FROM_BELOW ? &rbh[1] : nullptr);
                    pb = predict8(rb);
                }
                v2u pf = usef ? pred : pb;
                pb = useb ? pb : pred;
                pred.x = __builtin_amdgcn_lerp(pf.x, pb.x, 0x01010101u);
                pred.y = __builtin_amdgcn_lerp(pf.y, pb.y, 0x01010101u);
            }
            if (nopred) pred = v2u{0u, 0u};
            // clamp(t/256 + pred) == sat_u8((t + pred*256) >> 8)
            t[0] += pred_x256<0>(pred.x);
            t[1] += pred_x256<1>(pred.x);
            t[2] += pred_x256<2>(pred.x);
            t[3] += pred_x256<3>(pred.x);
            t[4] += pred_x256<0>(pred.y);
            t[5] += pred_x256<1>(pred.y);
            t[6] += pred_x256<2>(pred.y);
            t[7] += pred_x256<3>(pred.y);
        }
        v2u o;
        o.x = sat_pk4<8>(t[0], t[1], t[2], t[3]);
        o.y = sat_pk4<8>(t[4], t[5], t[6], t[7]);
        if constexpr (!DISPLAY) {
            __builtin_amdgcn_raw_buffer_store_b64(o, buf_rsrc(pd.out + plane_off), (int)out_voff, (int)(half ? half_step : 0u), 0);
        } else {
            // planes only for pictures that will be predicted from.  A scalar branch (the flag is the picture's): until round 4 the
            // store was issued into a resource without records, which drops it -- after the texture path has processed it; a B task
            // issued six such stores among its 54 memory instructions
#if LEON_NO_PLANES_BRANCH
            if (!pd.no_planes) __builtin_amdgcn_raw_buffer_store_b64(o, buf_rsrc(pd.out + plane_off), (int)out_voff, (int)(half ? half_step : 0u), 0);
#else
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(pd.out + plane_off), 0, pd.no_planes ? 0 : 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(o, prs, (int)out_voff, (int)(half ? half_step : 0u), 0);
#endif
            if constexpr (CHROMA) {
                // park the samples for the luma parts: [plane = half][row hi3][8 bytes of macroblock lo3]
                *reinterpret_cast<v2u*>(dsp.stash + half * 512 + hi3 * 64 + lo3 * 8) = o;
            } else if constexpr (AMODE == 1) {
                // yuva, A part: the samples wait for the Y part of the same macroblocks
                *reinterpret_cast<v2u*>(dsp.apark + half * 512 + hi3 * 64 + lo3 * 8) = o;
            } else {
                // converted right away (stage 5): the rows change lanes through the park
                if constexpr (Lay<LAYOUT>::park_in_tile) {      // the tile half is read out (stage 3): every lane's read has completed
                    wait_lds_all();
                    __builtin_amdgcn_wave_barrier();
                }
                *reinterpret_cast<v2u*>((Lay<LAYOUT>::park_in_tile ? tile + half * kLdsHalf : lds + kOffYpark) + hi3 * 64 + lo3 * 8) = o;
            }
        }
        if constexpr (DISPLAY && !CHROMA && AMODE != 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!(LEON_ABL & 4)) display_half<AMODE>(pd, G, dsp, Lay<LAYOUT>::park_in_tile ? tile + half * kLdsHalf : lds + kOffYpark, half, Rt, g, hi3, lo3);
            // the next half parks its rows in the same place
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ---- the back half of a luma part of recon_luma_pair as two functions (LEON_EARLY_RIGHT) -----------------------------------------
// What recon_task<TYPE, false, false, true, 0, 2, true, 1> does, cut in two at the point where the reference rows have been requested:
// luma_back_fetch (the macroblock's flags and vectors from the chroma part's lanes, window geometry, the requests) and luma_back_rest
// (row passes, prediction, stores, conversion).  Between a wave's request for reference rows and its first use of them lay one row
// pass (~170 vector instructions); with the cut, the RIGHT part's rows are requested while the left part still has its last half to
// convert and store (`hook`, called when the left part's own reference rows are dead: the registers are free), so they fly behind
// ~310 instructions and two store instructions more.
#ifndef LEON_EARLY_RIGHT
#define LEON_EARLY_RIGHT 0
#endif
struct NoHook { __device__ __forceinline__ void operator()() const {} };
struct LumaBack {
    uint32_t out_voff;
    int ovA, ovB;
    bool nopred, usef, useb, any_f, any_b;
    RefRows rfh[2], rbh[2];
};

template <int TYPE>
__device__ __forceinline__ void luma_back_fetch(LumaBack& S, const PicDesc& pd, const Geom& G, int Rt, int g, const char* lds, int lane, int side,
                                                const MbCarry& carry)
{
    const int W = G.cw, H = G.ch, bw = W >> 3;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int Qb = g * 8 + lo3;
    const bool valid = Qb < bw;
    const int Qs = valid ? Qb : bw - 1;
    // this lane's macroblock is number 4 * side + (block >> 1) of the task: its chroma-part lane holds it
    const int src = (4 * side + (lo3 >> 1)) << 2;
    const uint32_t flags = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)carry.flags);
    uint32_t mf = *reinterpret_cast<const uint32_t*>(lds + Lay<1>::carry + src), mk = 0;
    if (TYPE == 3) mk = *reinterpret_cast<const uint32_t*>(lds + Lay<1>::carry + 32 + src);
    const int x0 = 8 * Qs;
    bool nopred = (flags & 512u) != 0u, usef = true, useb = true;
    if (TYPE == 3) {
        const int dir = (int)(flags >> 10) & 3;
        usef = (dir & 1) != 0;
        useb = (dir & 2) != 0;
        nopred = nopred || dir == 0;
        if (!usef) mf = 0;
        if (!useb) mk = 0;
    }
    const int fh = (int)(short)(mf & 0xffff), fv = (int)mf >> 16;
    const int bh = (int)(short)(mk & 0xffff), bv = (int)mk >> 16;
    const int pxA = x0 + (fh >> 1), ohA = fh & 1, ayA = fv >> 1, ovA = fv & 1;
    bool inA = (uint32_t)pxA < (uint32_t)(W - 7 - ohA), inB = true;
    int pxB = 0, ohB = 0, ayB = 0, ovB = 0;
    if (TYPE == 3) {
        pxB = x0 + (bh >> 1); ohB = bh & 1; ayB = bv >> 1; ovB = bv & 1;
        inB = (uint32_t)pxB < (uint32_t)(W - 7 - ohB);
    }
    const bool useA = usef && !nopred, useB = useb && !nopred;
    inA = inA || !useA;
    inB = inB || !useB;
    S.out_voff = ((uint32_t)__mul24(16 * Rt + hi3, W) + (uint32_t)x0) | (valid ? 0u : kOobBit);
    S.ovA = ovA; S.ovB = ovB; S.nopred = nopred; S.usef = usef; S.useb = useb;
    S.any_f = TYPE != 3 || __builtin_amdgcn_ballot_w64(useA) != 0;
    S.any_b = TYPE == 3 && __builtin_amdgcn_ballot_w64(useB) != 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const bool ninth = hi3 == 7 && (h == 1 || !LEON_NINTH_FROM_BELOW);
        if (S.any_f) S.rfh[h] = fetch_rows(gptr(pd.ref_fwd), W, H, 8 * (2 * Rt + h) + hi3, pxA, ayA, ohA, ovA, inA, ninth, useA);
        if (TYPE == 3 && S.any_b) S.rbh[h] = fetch_rows(gptr(pd.ref_bwd), W, H, 8 * (2 * Rt + h) + hi3, pxB, ayB, ohB, ovB, inB, ninth, useB);
    }
}

template <int TYPE, typename HOOK>
__device__ __forceinline__ void luma_back_rest(LumaBack& S, const PicDesc& pd, const Geom& G, int Rt, int g, int lane, const Display& dsp, char* tile,
                                               uint64_t live0, uint64_t live1, HOOK hook)
{
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const uint32_t half_step = 8u * (uint32_t)G.cw;
    const uint64_t live[2] = {live0, live1};
#pragma unroll
    for (int half = 0; half < 2; half++) {
        int t[8];
        const uint64_t colbits = live[half];
        if (colbits == 0) {
#pragma unroll
            for (int m = 0; m < 8; m++) t[m] = 128;
        } else {
            // stage 3, the row pass (recon_task)
            const int cols_live = 8 - (__builtin_clzll(colbits | 1ull) >> 3);
            const v4u wv = *reinterpret_cast<const v4u*>(tile + half * kLdsHalf + hi3 * 128 + lo3 * 16);
            const v2f k25 = {2.5f, 2.5f};
            const v2f a = v2f{(float)(short)(wv.x & 0xffffu), (float)((int)wv.x >> 16)} * k25;
            const int Y0 = (int)a.x + 128, Y1 = (int)a.y;
            if (cols_live <= 2) {
                butterfly8_lo2(Y0, Y1, t);
            } else {
                const v2f bb = v2f{(float)(short)(wv.y & 0xffffu), (float)((int)wv.y >> 16)} * k25;
                if (cols_live <= 4) {
                    butterfly8_lo4(Y0, Y1, (int)bb.x, (int)bb.y, t);
                } else {
                    const v2f cc = v2f{(float)(short)(wv.z & 0xffffu), (float)((int)wv.z >> 16)} * k25;
                    const v2f dd = v2f{(float)(short)(wv.w & 0xffffu), (float)((int)wv.w >> 16)} * k25;
                    const int Y[8] = {Y0, Y1, (int)bb.x, (int)bb.y, (int)cc.x, (int)cc.y, (int)dd.x, (int)dd.y};
                    butterfly8(Y, t);
                }
            }
#pragma unroll
            for (int m = 0; m < 8; m++) t[m] += (t[m] >> 31) & 255;
        }
        // stage 4: prediction, add, clamp, store
        v2u pred = {0u, 0u};
        if (S.any_f) {
            finish_rows(S.rfh[half], S.ovA, hi3 == 7, lane, half == 0 && LEON_NINTH_FROM_BELOW ? &S.rfh[1] : nullptr);
            pred = predict8(S.rfh[half]);
        }
        if (TYPE == 3) {
            v2u pb = {0u, 0u};
            if (S.any_b) {
                finish_rows(S.rbh[half], S.ovB, hi3 == 7, lane, half == 0 && LEON_NINTH_FROM_BELOW ? &S.rbh[1] : nullptr);
                pb = predict8(S.rbh[half]);
            }
            const v2u pf = S.usef ? pred : pb;
            pb = S.useb ? pb : pred;
            pred.x = __builtin_amdgcn_lerp(pf.x, pb.x, 0x01010101u);
            pred.y = __builtin_amdgcn_lerp(pf.y, pb.y, 0x01010101u);
        }
        if (S.nopred) pred = v2u{0u, 0u};
        t[0] += pred_x256<0>(pred.x);
        t[1] += pred_x256<1>(pred.x);
        t[2] += pred_x256<2>(pred.x);
        t[3] += pred_x256<3>(pred.x);
        t[4] += pred_x256<0>(pred.y);
        t[5] += pred_x256<1>(pred.y);
        t[6] += pred_x256<2>(pred.y);
        t[7] += pred_x256<3>(pred.y);
        v2u o;
        o.x = sat_pk4<8>(t[0], t[1], t[2], t[3]);
        o.y = sat_pk4<8>(t[4], t[5], t[6], t[7]);
        // the part's own reference rows are dead: the next part's may be requested
        if (half == 1) hook();
        if (!pd.no_planes) __builtin_amdgcn_raw_buffer_store_b64(o, buf_rsrc(pd.out), (int)S.out_voff, (int)(half ? half_step : 0u), 0);
        // the tile half is read out (stage 3): every lane's read has completed; the rows change lanes through it
        wait_lds_all();
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v2u*>(tile + half * kLdsHalf + hi3 * 64 + lo3 * 8) = o;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!(LEON_ABL & 4)) display_half<0>(pd, G, dsp, tile + half * kLdsHalf, half, Rt, g, hi3, lo3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The two luma parts of a dense display task (P and B pictures) with ONE front: the coefficient rows of both parts are requested
// together (two tiles), one liveness scan lists the live columns of all four halves, ONE column pass takes them -- a P or B part
// has about a dozen live columns of 128, and a pass costs ~140 instructions however few of its 64 lanes have a column -- and then each
// part runs its back half (maps from the chroma part, reference fetches, row passes, prediction, stores, conversion) as before.
// I pictures keep a front per part: with ~40 live columns per part the shared pass would run twice anyway.
template <int TYPE>
__device__ __forceinline__ void recon_luma_pair(const PicDesc& pd, const Geom& G, int Rt, int gc, char* lds, int lane, Display dsp, MbCarry& carry, bool has_right,
                                                uint32_t qreg)
{
    const int W = G.cw, hi3 = lane >> 3, lo3 = lane & 7;
    char* const tileR = lds + kOffTileR;
    // the previous part of this wave (the chroma part) may still be reading the tile
    wait_lds_all();
    __builtin_amdgcn_wave_barrier();
    const uint32_t row_off = (uint32_t)__mul24(16 * Rt + hi3, W);
    {
        const int Qld = 16 * gc + lo3;
        const uint32_t voff = (2u * (row_off + (uint32_t)(8 * Qld))) | (Qld < (W >> 3) ? 0u : kOobBit);
        coef_rows_to_lds(pd.coef[0], lds, voff, 0u);
        coef_rows_to_lds(pd.coef[0], lds + kLdsHalf, voff, 16u * (uint32_t)W);
    }
    if (has_right) {
        const int Qld = 16 * gc + 8 + lo3;
        const uint32_t voff = (2u * (row_off + (uint32_t)(8 * Qld))) | (Qld < (W >> 3) ? 0u : kOobBit);
        coef_rows_to_lds(pd.coef[0], tileR, voff, 0u);
        coef_rows_to_lds(pd.coef[0], tileR + kLdsHalf, voff, 16u * (uint32_t)W);
    }
#if LEON_EARLY_RIGHT
    // P pictures (52 registers, five waves per SIMD by LDS: room for 96): the reference rows of BOTH parts are requested right behind
    // the coefficient rows -- one trip to memory for everything the luma parts read
    constexpr bool kRefsUpFront = LEON_EARLY_RIGHT >= 2 && TYPE == 2;
    constexpr bool kLeftUpFront = kRefsUpFront || (LEON_EARLY_RIGHT == 4 && TYPE == 3);     // (B: the left part's only; 71 registers)
    LumaBack backL, backR;
    if (kLeftUpFront) luma_back_fetch<TYPE>(backL, pd, G, Rt, 2 * gc, lds, lane, 0, carry);
    if (kRefsUpFront && has_right) luma_back_fetch<TYPE>(backR, pd, G, Rt, 2 * gc + 1, lds, lane, 1, carry);
#endif
    if (LEON_PRIO_OF(TYPE) == 3) __builtin_amdgcn_s_setprio(0);
    // quantiser scale | intra << 8 of the macroblock of block b of either part, in lane b (the chroma part's lanes hold the task's
    // eight macroblocks: lane m, macroblock m)
    const int qiaL = __builtin_amdgcn_ds_bpermute((lo3 >> 1) << 2, (int)carry.flags) & 0x11f;
    const int qiaR = __builtin_amdgcn_ds_bpermute((4 + (lo3 >> 1)) << 2, (int)carry.flags) & 0x11f;
    wait_vmem_all();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if LEON_EARLY_RIGHT
    // B pictures (70 registers of 72): the LEFT part's reference rows are requested as soon as the coefficient rows are in (a full
    // wait in front of them would wait for these too) -- the scan and the column pass run while they fly
    constexpr bool kLeftBeforeScan = LEON_EARLY_RIGHT == 3 && TYPE == 3;
    if (kLeftBeforeScan) luma_back_fetch<TYPE>(backL, pd, G, Rt, 2 * gc, lds, lane, 0, carry);
#endif
    uint64_t liveL[2], liveR[2] = {0, 0};
    uint32_t n_cols = scan_tile(lds, lds + Lay<1>::slots, lane, 0u, 0u, liveL);
    if (has_right) n_cols = scan_tile(tileR, lds + Lay<1>::slots, lane, 128u, n_cols, liveR);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    column_pass<true, false>(lds, (uint32_t)kOffTileR, lds + Lay<1>::slots, nullptr, n_cols, qiaL, qiaR, lane, qreg);
    if (LEON_PRIO_OF(TYPE) == 4) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    dsp.side = 0;
#if LEON_EARLY_RIGHT
    {
        if (!kLeftUpFront && !kLeftBeforeScan) luma_back_fetch<TYPE>(backL, pd, G, Rt, 2 * gc, lds, lane, 0, carry);
        luma_back_rest<TYPE>(backL, pd, G, Rt, 2 * gc, lane, dsp, lds, liveL[0], liveL[1],
                             [&]() { if (!kRefsUpFront && has_right) luma_back_fetch<TYPE>(backR, pd, G, Rt, 2 * gc + 1, lds, lane, 1, carry); });
        if (has_right) {
            dsp.side = 1;
            luma_back_rest<TYPE>(backR, pd, G, Rt, 2 * gc + 1, lane, dsp, tileR, liveR[0], liveR[1], NoHook{});
        }
        return;
    }
#endif
    recon_task<TYPE, false, false, true, 0, 2, true, 1>(pd, G, Rt, 2 * gc, lds, lane, dsp, carry, false, lds, liveL[0], liveL[1]);
    if (has_right) {
        if (LEON_PRIO_OF(TYPE) == 1) __builtin_amdgcn_s_setprio(3);
        if (LEON_PRIO_OF(TYPE) == 5) __builtin_amdgcn_s_setprio(0);
        dsp.side = 1;
        recon_task<TYPE, false, false, true, 0, 2, true, 1>(pd, G, Rt, 2 * gc + 1, lds, lane, dsp, carry, false, tileR, liveR[0], liveR[1]);
    }
}

// A wave's copy of the quantiser matrices and the premultiplier (the first 64 dwords of Tables; the column pass
// reads them by the column its lane was handed): one load and one LDS write per lane, once per wave.
// Straight into LDS (lane i's dword at kOffQtab + 4 i), nothing waits here: through a register (round 1-4: load, s_waitcnt vmcnt(0),
// ds_write) the wave made a round trip to memory before it requested its first coefficient row.  The tables are read by the column
// pass, behind the task's wait for its coefficient rows (dense: wait_vmem_all) or first entries (sparse; loads return in order).
__device__ __forceinline__ void stage_tables(const QTables* __restrict__ Tg, char* lds, int lane)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Tg, 0, 256, 0x00020000);
    // (the matrices and the premultiplier only: the 64 bytes behind them are the display task's vectors in LDS, kOffCarry, written
    // by the chroma part -- which a load that lands late must not overwrite)
    if (lane < 48) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + kOffQtab), 4, lane * 4, 0, 0, 0);
}

// XCD-aware workgroup remap: hardware deals workgroups round-robin over the 8 XCDs;
// give each XCD one contiguous run of the task space so that vertically adjacent
// block groups (which share reference lines) meet in the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
    int q = n >> 3, r = n & 7;
    int x = bid & 7, j = bid >> 3;
    return x * q + min(x, r) + j;
}

// workgroup -> (picture, workgroup inside the picture).  B pictures come in pairs that predict from the same two
// anchors (the host keeps the pictures of a GOP adjacent): the workgroups of two consecutive pictures alternate, so
// both read the same reference lines at about the same time and the second read finds them in the XCD's L2 -- one
// after the other, a reference pair (6 MB at 1080p) is gone from the 4 MB L2 before it is used again.
template <int TYPE>
__device__ __forceinline__ void pic_of_wg(const Geom& G, int wg, int& pic, int& twg)
{
    if (TYPE == 3 && LEON_PAIR_B) {
        const int half = wg >> 1;
        const int pair = div_inv(half, G.inv_wg_per_pic);
        twg = half - pair * G.wg_per_pic;
        pic = 2 * pair + (wg & 1);
    } else {
        pic = div_inv(wg, G.inv_wg_per_pic);
        twg = wg - pic * G.wg_per_pic;
    }
}

template <int TYPE, bool SPARSE>
__device__ __forceinline__ void recon_dispatch(const PicDesc& pd, const Geom& G, int t, char* lds, int lane)
{
    const Display none{nullptr, 0};
    MbCarry own{};
    if (t < G.tasksY) {
        int Rt = div_inv(t, G.inv_gY), g = t - Rt * G.gY;
        recon_task<TYPE, false, SPARSE, false>(pd, G, Rt, g, lds, lane, none, own);
    } else if (t < G.tasksY + G.tasksC) {
        t -= G.tasksY;
        int Rt = div_inv(t, G.inv_gC), g = t - Rt * G.gC;
        recon_task<TYPE, true, SPARSE, false>(pd, G, Rt, g, lds, lane, none, own);
    } else {                                   // yuva: the A plane, luma-shaped
        t -= G.tasksY + G.tasksC;
        int Rt = div_inv(t, G.inv_gY), g = t - Rt * G.gY;
        recon_task<TYPE, false, SPARSE, false>(pd, G, Rt, g, lds, lane, none, own, true);
    }
}

// One kernel per picture type: the register budget of the I and P paths is not held hostage
// by the two predictors of the B path (VGPRs decide waves per SIMD), and a launch only ever
// contains pictures of one type.
template <int TYPE, bool SPARSE>
__global__ __launch_bounds__(kReconMaxThreads) void k_recon(const PicDesc* __restrict__ descs, Geom G,
                                               const Tables* __restrict__ T)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wg = xcd_remap(blockIdx.x, G.n_wg);
    int pic, twg;
    pic_of_wg<TYPE>(G, wg, pic, twg);
    const int t = twg * kWavesPerWG + wave;
    if (t >= G.tasks_per_pic || pic >= G.n_pics) return;
    char* lds = smem + wave * kLdsPerWave;
    const PicDesc& pd = descs[pic];
    stage_tables(pd.qt, lds, lane);
    recon_dispatch<TYPE, SPARSE>(pd, G, t, lds, lane);
}

// The same reconstruction with the display conversion fused in (see Display above).  One wave = the 8
// macroblocks of one chroma group, all components: tasks_per_pic = tasksC here (the host sets Geom up
// for that), the chroma part first, then the left and the right luma part.
// The sparse B kernel (what the pipeline runs most) is held to 72 registers = 7 waves per SIMD: it wants 74, the two
// spilled dwords (8 bytes of scratch per lane) cost less than the wave brings: +1..2 % end to end, three pairs on one box.
// one display task of a wave: the chroma part, the workgroup's barrier in front of the first table lookup (`first`: every wave of the
// workgroup comes by here exactly once), the luma parts.  false: the wave has no task (and none behind this one).
template <int TYPE, bool SPARSE, bool ALPHA>
__device__ __forceinline__ bool display_task(const PicDesc* __restrict__ descs, const Geom& G, int pic, int t, char* lds, int lane, const char* lut, bool first)
{
    constexpr bool kPair = LEON_PAIR_LUMA && LEON_CARRY && !SPARSE && !ALPHA && TYPE != 1;      // recon_luma_pair
    const bool live = t < G.tasks_per_pic && pic < G.n_pics;
    const PicDesc& pd = descs[live ? pic : 0];
    const int Rt = div_inv(t, G.inv_gC), gc = t - Rt * G.gC;
    Display dsp{lds + (kPair ? Lay<1>::stash : Lay<0>::stash), 0, lds + kLdsPerWaveDisplay, lut};
    if constexpr (!kPair) {
        if (!first) { wait_lds_all(); __builtin_amdgcn_wave_barrier(); }      // the column pass of the task before has read its tables
        stage_tables(pd.qt, lds, lane);
    }
    MbCarry carry{};
    // the two-tile layout has no room for the quantiser tables in LDS: the wave keeps them in ONE register, dword i of the 256 bytes
    // in lane i (column_pass takes what it needs by ds_bpermute); requested first, landed with the macroblock maps
    uint32_t qreg = 0u;
    if constexpr (kPair) qreg = ldg<uint32_t>(gptr(pd.qt), (uint32_t)lane * 4u);
    if (live) recon_task<TYPE, true, SPARSE, true, 0, LEON_CARRY ? 1 : 0, false, kPair ? 1 : 0>(pd, G, Rt, gc, lds, lane, dsp, carry, false, nullptr, 0, 0, qreg);
    if (first) {
        if (!live) wait_vmem_all();      // (a wave with a task has waited for memory behind its chroma part's loads: its chunks of the tables are in)
        __syncthreads();                 // the conversion tables have landed: every wave's chunks
    }
    if (!live) return false;
    if (kPair && LEON_PRIO_OF(TYPE) == 1) __builtin_amdgcn_s_setprio(1);
    if (kPair && LEON_PRIO_OF(TYPE) == 2) __builtin_amdgcn_s_setprio(0);
    // the two luma parts as two calls, not a loop: the loop form keeps 15 more registers live (B path: 93).
    // yuva: the A part of the same four macroblocks first (AMODE 1), then the Y part that displays them (AMODE 2).
    if constexpr (kPair) {
        recon_luma_pair<TYPE>(pd, G, Rt, gc, lds, lane, dsp, carry, 2 * gc + 1 < G.gY, qreg);
        return true;
    }
    dsp.side = 0;
    if constexpr (ALPHA) {
        recon_task<TYPE, false, SPARSE, true, 1, LEON_CARRY ? 2 : 0>(pd, G, Rt, 2 * gc, lds, lane, dsp, carry, true);
        recon_task<TYPE, false, SPARSE, true, 2, LEON_CARRY ? 2 : 0>(pd, G, Rt, 2 * gc, lds, lane, dsp, carry);
    } else {
        recon_task<TYPE, false, SPARSE, true, 0, LEON_CARRY ? 2 : 0>(pd, G, Rt, 2 * gc, lds, lane, dsp, carry);
    }
    if (2 * gc + 1 < G.gY) {
        dsp.side = 1;
        if constexpr (ALPHA) {
            recon_task<TYPE, false, SPARSE, true, 1, LEON_CARRY ? 2 : 0>(pd, G, Rt, 2 * gc + 1, lds, lane, dsp, carry, true);
            recon_task<TYPE, false, SPARSE, true, 2, LEON_CARRY ? 2 : 0>(pd, G, Rt, 2 * gc + 1, lds, lane, dsp, carry);
        } else {
            recon_task<TYPE, false, SPARSE, true, 0, LEON_CARRY ? 2 : 0>(pd, G, Rt, 2 * gc + 1, lds, lane, dsp, carry);
        }
    }
    return true;
}

template <int TYPE, bool SPARSE, bool ALPHA = false>
__global__ __launch_bounds__(kReconMaxThreads) __attribute__((amdgpu_waves_per_eu(TYPE == 3 && !ALPHA && (SPARSE || LEON_CARRY) ? 7 : 4)))
void k_recon_display(const PicDesc* __restrict__ descs, Geom G,
                                                                    const Tables* __restrict__ T)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (the dense, non-alpha kernels only: in the pipeline's sparse launches a raised priority takes issue slots from the parser
    // kernels beside them -- 175-177 k against 181-182 k pictures/s end to end)
    if ((LEON_PAIR_LUMA && LEON_CARRY && !SPARSE && !ALPHA && TYPE != 1) && LEON_PRIO_OF(TYPE) >= 2) __builtin_amdgcn_s_setprio(3);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    const int wg = xcd_remap(blockIdx.x, G.n_wg);
    int pic, twg;
    pic_of_wg<TYPE>(G, wg, pic, twg);
    const int wpw = (int)(blockDim.x >> 6);
    // the conversion tables: 5 KB per workgroup, requested before anything else and needed only after the
    // chroma part -- the barrier in display_task finds them long landed.  Straight into LDS, 1 KB per instruction (lane i's 16
    // bytes at chunk + 16 i), the chunks dealt to the workgroup's waves: nothing waits for them here.  (Until round 4 the
    // tables went through registers: load, s_waitcnt vmcnt(0), ds_write -- a round trip to memory at the head of EVERY
    // workgroup, in front of its first coefficient load; a workgroup of the B kernel lives 24 us.)  A wave's chunks have landed
    // when its next full wait for memory returns: the chroma part's, behind its coefficient loads (dense) or first entries
    // (sparse; loads return in order) -- a wave without a task waits in display_task -- and the barrier there makes every
    // wave's chunks everybody's.
    __shared__ __attribute__((aligned(16))) int32_t lut_s[kLdsLut / 4];      // static: its LDS address is a compile-time constant
    {
        static_assert(kLdsLut % 1024 == 0, "whole chunks");
        const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)T->rgba_lut, 0, kLdsLut, 0x00020000);
        const int n_waves = (int)(blockDim.x >> 6);
        for (int c = wave; c < kLdsLut / 1024; c += n_waves)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(lut_s) + c * 1024), 16,
                                                     (int)(lane0 * 16u), c * 1024, 0, 0);
    }
    constexpr bool kPair = LEON_PAIR_LUMA && LEON_CARRY && !SPARSE && !ALPHA && TYPE != 1;
    char* lds = smem + wave * (ALPHA ? kLdsPerWaveDisplayAlpha : (kPair ? kLdsPerWaveDisplayPair : kLdsPerWaveDisplay));
    // (Round 4 tried a wave running two to five tasks one after the other, so that the frames' stores of a task drain while the wave
    // works on the next: 5.89-5.92 ms per step with two against 5.90-5.95 with one, worse with three and five, and 10-20 registers more
    // for the loop -- not kept.)
    display_task<TYPE, SPARSE, ALPHA>(descs, G, pic, twg * wpw + wave, lds, lane0, reinterpret_cast<const char*>(lut_s), true);
}

// ---- K3: YCbCr 4:2:0 -> RGBA8 ------------------------------------------------------

struct RgbaGeom {
    int32_t cw, ch, fw, fh;
    int32_t cols, rows;          // fw>>1, fh>>1 quads
    int32_t n;                   // frames in the batch
    int32_t flavour;             // 0 CPU twin (fp64), 1 GL (fp32); bit 8: yuva -- the A byte comes from the slot's fourth plane
    uint32_t slot_stride_lo, slot_stride_hi;   // bytes between slots
    uint32_t inv_cols4;          // ceil(2^32 / (fw/4)): the 4x2 kernel walks a frame linearly
};

// Uint8ClampedArray store: clamp, round half to even (2^52+2^51 trick; |x| < 2^31)
__device__ __forceinline__ uint32_t u8_clamped(double x)
{
    double t = x + 6755399441055744.0;
    int i = __double2loint(t);
    return (uint32_t)med3i(i, 0, 255);
}

// CPU twin, one thread per 2x2 quad, with the reference's flat index progression
// (player/easybits.player.js:2692-2782): identical to a plain crop for even frame
// widths, and reproducing its one-sample-per-row-pair drift for odd ones.
__global__ __launch_bounds__(kRgbaBlock) void k_rgba_twin(const uint8_t* __restrict__ slots, const int32_t* __restrict__ slot_ids,
                                                   uint8_t* __restrict__ rgba, RgbaGeom G)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    const int f = blockIdx.z;
    if (col >= G.cols) return;
    const size_t stride = ((size_t)G.slot_stride_hi << 32) | G.slot_stride_lo;
    const uint8_t* Y = slots + (size_t)slot_ids[f] * stride;
    const uint8_t* Cb = Y + (size_t)G.cw * G.ch;
    const uint8_t* Cr = Cb + ((size_t)G.cw * G.ch >> 2);
    uint8_t* dst = rgba + (size_t)f * G.fw * G.fh * 4;
    const int odd = G.fw & 1;
    const int hw = G.cw >> 1;
    const int yi1 = row * (2 * G.cw - odd) + 2 * col;
    const int yi2 = yi1 + G.cw;
    const int ci = row * hw + col;
    const size_t d1 = 4 * ((size_t)row * (2 * G.fw - odd) + 2 * col);
    const size_t d2 = d1 + 4 * (size_t)G.fw;
    const double yuvr = (double)Cr[ci] - 128.0, yuvb = (double)Cb[ci] - 128.0;
    const double r = yuvr * 1.59603;
    const double g = (-0.81297 * yuvr) - (0.39176 * yuvb);
    const double b = yuvb * 2.01723;
    uint32_t px[4];
    const int yidx[4] = {yi1, yi1 + 1, yi2, yi2 + 1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double ys = ((double)Y[yidx[k]] - 16.0) * 1.16438;
        px[k] = u8_clamped(r + ys) | (u8_clamped(g + ys) << 8) | (u8_clamped(b + ys) << 16) | 0xff000000u;
    }
    *reinterpret_cast<uint32_t*>(dst + d1) = px[0];
    *reinterpret_cast<uint32_t*>(dst + d1 + 4) = px[1];
    *reinterpret_cast<uint32_t*>(dst + d2) = px[2];
    *reinterpret_cast<uint32_t*>(dst + d2 + 4) = px[3];
}

// Fast path of the CPU twin for frame widths that are a multiple of 4 (no index drift): one thread =
// 4 x 2 pixels = two 2x2 quads, same fp64 operations in the same order as k_rgba_twin.  The 64 lanes
// of a wave store 1 KB of contiguous RGBA per row with one 16-byte store each -- 5.8 TB/s; the
// 8 x 2 form (two 16-byte stores per lane and row, lanes 32 bytes apart) reached 5.4.
__global__ __launch_bounds__(kRgbaBlock) void k_rgba_twin4(const uint8_t* __restrict__ slots, const int32_t* __restrict__ slot_ids,
                                                    uint8_t* __restrict__ rgba, RgbaGeom G)
{
    // one thread per (row pair, group of 4 columns), numbered linearly through the frame: a 1920-wide
    // row pair is 7.5 waves, and a row-shaped grid would leave every eighth wave half empty
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cols4 = (uint32_t)G.fw >> 2;
    if (idx >= cols4 * (uint32_t)G.rows) return;
    const int row = (int)(G.inv_cols4 ? __umulhi(idx, G.inv_cols4) : idx);   // exact: idx * cols4 < 2^32
    const int col4 = (int)(idx - (uint32_t)row * cols4);
    const int f = blockIdx.z;
    const size_t stride = ((size_t)G.slot_stride_hi << 32) | G.slot_stride_lo;
    const uint8_t* Y = slots + (size_t)slot_ids[f] * stride;
    const uint8_t* Cb = Y + (size_t)G.cw * G.ch;
    const uint8_t* Cr = Cb + ((size_t)G.cw * G.ch >> 2);
    const int hw = G.cw >> 1;
    const uint32_t y0 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(Y + (size_t)(2 * row) * G.cw + 4 * col4));
    const uint32_t y1 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(Y + (size_t)(2 * row + 1) * G.cw + 4 * col4));
    const uint32_t cb2 = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(Cb + (size_t)row * hw + 2 * col4));
    const uint32_t cr2 = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(Cr + (size_t)row * hw + 2 * col4));
    uint32_t o0[4], o1[4];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const double yuvr = (double)((cr2 >> (8 * q)) & 255u) - 128.0, yuvb = (double)((cb2 >> (8 * q)) & 255u) - 128.0;
        const double r = yuvr * 1.59603;
        const double g = (-0.81297 * yuvr) - (0.39176 * yuvb);
        const double b = yuvb * 2.01723;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int px = 2 * q + k;
            const double ya = ((double)((y0 >> (8 * px)) & 255u) - 16.0) * 1.16438;
            const double yb = ((double)((y1 >> (8 * px)) & 255u) - 16.0) * 1.16438;
            o0[px] = u8_clamped(r + ya) | (u8_clamped(g + ya) << 8) | (u8_clamped(b + ya) << 16) | 0xff000000u;
            o1[px] = u8_clamped(r + yb) | (u8_clamped(g + yb) << 8) | (u8_clamped(b + yb) << 16) | 0xff000000u;
        }
    }
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(rgba + (size_t)f * G.fw * G.fh * 4);
    const uint32_t o = ((uint32_t)(2 * row) * (uint32_t)G.fw + 4u * (uint32_t)col4) * 4u;
    __builtin_amdgcn_raw_buffer_store_b128(v4u{o0[0], o0[1], o0[2], o0[3]}, rs, (int)o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v4u{o1[0], o1[1], o1[2], o1[3]}, rs, (int)(o + (uint32_t)G.fw * 4u), 0, 0);
}

// yuva: the A byte of every converted pixel comes from the slot's fourth plane (behind Cr) instead of the
// constant 255 -- a second pass over the frame, launched for alpha decoders only so that the conversion
// kernels above stay as they are.  One thread per pixel quad of a row (frame width is even for alpha).
__global__ __launch_bounds__(kRgbaBlock) void k_rgba_alpha(const uint8_t* __restrict__ slots, const int32_t* __restrict__ slot_ids,
                                                           uint8_t* __restrict__ rgba, RgbaGeom G, int cover_w, int cover_h)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int yy = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= cover_w || yy >= cover_h) return;
    const size_t stride = ((size_t)G.slot_stride_hi << 32) | G.slot_stride_lo;
    const size_t ysz = (size_t)G.cw * G.ch;
    const uint8_t* A = slots + (size_t)slot_ids[f] * stride + ysz + (ysz >> 1);
    uint8_t* dst = rgba + ((size_t)f * G.fw * G.fh + (size_t)yy * G.fw + x) * 4;
    dst[3] = A[(size_t)yy * G.cw + x];
    if (x + 1 < cover_w) dst[7] = A[(size_t)yy * G.cw + x + 1];
}

// fills what the quad loop never writes (odd last row / column, drift leftovers) with 255
__global__ __launch_bounds__(kRgbaBlock) void k_fill255(uint32_t* __restrict__ p, size_t n_dwords)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_dwords) p[i] = 0xffffffffu;
}

// GL flavour: every pixel of the frame_w x frame_h crop, fp32, left-to-right, no contraction
__global__ __launch_bounds__(kRgbaBlock) void k_rgba_gl(const uint8_t* __restrict__ slots, const int32_t* __restrict__ slot_ids,
                                                 uint8_t* __restrict__ rgba, RgbaGeom G)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= G.fw) return;
    const size_t stride = ((size_t)G.slot_stride_hi << 32) | G.slot_stride_lo;
    const uint8_t* Y = slots + (size_t)slot_ids[f] * stride;
    const uint8_t* Cb = Y + (size_t)G.cw * G.ch;
    const uint8_t* Cr = Cb + ((size_t)G.cw * G.ch >> 2);
    const int hw = G.cw >> 1;
    const float fy = (float)Y[(size_t)yy * G.cw + x] / 255.0f;
    const float fcb = (float)Cb[(size_t)(yy >> 1) * hw + (x >> 1)] / 255.0f;
    const float fcr = (float)Cr[(size_t)(yy >> 1) * hw + (x >> 1)] / 255.0f;
    const float M[3][4] = {{1.16438f, 0.00000f, 1.59603f, -0.87079f},
                           {1.16438f, -0.39176f, -0.81297f, 0.52959f},
                           {1.16438f, 2.01723f, 0.00000f, -1.08139f}};
    uint32_t o = 0xff000000u;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float s = fy * M[c][0];
        s = s + fcb * M[c][1];
        s = s + fcr * M[c][2];
        s = s + M[c][3];
        s = fminf(fmaxf(s, 0.0f), 1.0f);
        o |= (uint32_t)__float2int_rn(s * 255.0f) << (8 * c);
    }
    *reinterpret_cast<uint32_t*>(rgba + ((size_t)f * G.fw * G.fh + (size_t)yy * G.fw + x) * 4) = o;
}

// ---- measured HBM roofline -----------------------------------------------------------
// One 16-byte element per thread, no loop: the fastest of the copy shapes probed on MI355X
// (tools/probe/bw_probe.cpp: 6.3 TB/s vs 4.8-5.9 TB/s for grid-stride forms).
__global__ __launch_bounds__(kRgbaBlock) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// The same shape with R source streams and W destination streams (16 B per lane and stream, no loop): the yardstick for
// launches that do not read and write in equal parts -- an I launch of the fused path writes 55-65 % of its bytes.
// R = 0: write only; W = 0: read only (the store is there for the compiler and never happens).
template <int R, int W>
__global__ __launch_bounds__(kRgbaBlock) void k_stream16(const uint4* __restrict__ s0, const uint4* __restrict__ s1,
                                                         uint4* __restrict__ d0, uint4* __restrict__ d1, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 v = uint4{(uint32_t)i, 0x5a5a5a5au, (uint32_t)(i >> 32), 0xa5a5a5a5u};
    if (R >= 1) v = s0[i];
    if (R >= 2) { const uint4 u = s1[i]; v.x ^= u.x; v.y ^= u.y; v.z ^= u.z; v.w ^= u.w; }
    if (W >= 1) d0[i] = v;
    if (W >= 2) d1[i] = v;
    if (W == 0 && v.x == 0x12345678u && v.y == 0x9abcdef0u && v.z == 0x0fedcba9u) d0[i] = v;      // never (sources hold 0x5a bytes)
}

}  // namespace leon
