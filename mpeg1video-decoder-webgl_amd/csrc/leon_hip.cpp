// leon_hip.cpp -- the C ABI of include/leon.h over HIP (gfx950).
// Compiled with hipcc -x hip; see mpeg1video-decoder-webgl_amd/csrc/Makefile.
// There is no CPU path in this library: without a device every call fails.
#include "leon_kernels.h"
#include "leon_rgba_lut.h"
#include "../../include/leon.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace leon;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(LEON_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// decoders/jsv.js:1777-1806 default tables (same data as the oracle's, stated here
// because the product never links the oracle)
const uint8_t kDefaultIntra[64] = {
     8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};
const uint8_t kPremultiplier[64] = {
    32, 44, 42, 38, 32, 25, 17,  9, 44, 62, 58, 52, 44, 35, 24, 12,
    42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
    32, 44, 42, 38, 32, 25, 17,  9, 25, 35, 33, 30, 25, 20, 14,  7,
    17, 24, 23, 20, 17, 14,  9,  5,  9, 12, 12, 10,  9,  7,  5,  2};

struct TimedLaunch {
    hipEvent_t a, b;
    int kind;
    int pic_type;      // reconstruction launches: LEON_PIC_I/P/B, else 0
    double bytes;
    uint64_t mbs;
};

// SURVEY.md 8d / BASELINE.md section 2: algorithmic bytes per macroblock -- priced by what the
// macroblock itself must move: 768 B of coefficients + 384 B written + its map bytes and vectors,
// and 384 B per reference it really predicts from.  I 1154; P 1158 + 384 when predicted (= the
// survey's 1542; an intra macroblock of a P picture reads no reference); B 1162 + 384 per direction
// used (1546 one-sided, 1930 bidirectional = the survey's figure).
const double kBytesI = 1154.0, kBytesPBase = 1158.0, kBytesBBase = 1162.0, kBytesRef = 384.0;
const double kRgbaBytesPerMb = 1408.0;

// dense or sparse picture as the internals see it
struct AnyPic {
    leon_picture p{};            // coef_* unused when sparse
    const uint32_t* grp_off = nullptr;
    const uint32_t* entries = nullptr;
    uint32_t n_entries = 0;
    bool sparse = false;
};
AnyPic any_of(const leon_picture& p)
{
    AnyPic a;
    a.p = p;
    return a;
}
AnyPic any_of(const leon_sparse_picture& q)
{
    AnyPic a;
    a.p.type = q.type; a.p.out_slot = q.out_slot; a.p.ref_fwd_slot = q.ref_fwd_slot; a.p.ref_bwd_slot = q.ref_bwd_slot;
    a.p.qscale = q.qscale; a.p.intra = q.intra; a.p.repadd = q.repadd;
    a.p.mv_fwd = q.mv_fwd; a.p.mv_bwd = q.mv_bwd; a.p.mb_dir = q.mb_dir;
    a.p.rgba_out = q.rgba_out; a.p.no_planes = q.no_planes; a.p.qm_set = q.qm_set;
    a.grp_off = q.grp_off; a.entries = q.entries; a.n_entries = q.n_entries;
    a.sparse = true;
    return a;
}

struct Staging {                 // device copy of one host-submitted picture
    char* base = nullptr;
    char* host = nullptr;        // pinned mirror: the caller's arrays are gathered here, then ONE async copy
    hipEvent_t done = nullptr;
    bool busy = false;
};

}  // namespace

struct leon_batch {
    // descriptors sorted by launch class: class = type - 1 (planes only) or 3 + type - 1 (fused display conversion)
    PicDesc* d_descs = nullptr;
    int n = 0;
    int count[6] = {0, 0, 0, 0, 0, 0};
    std::vector<int32_t> out_slots;
    bool sparse = false;
    uint64_t entries_of_type[6] = {0, 0, 0, 0, 0, 0};   // sparse: list lengths per class (algorithmic bytes)
    double bytes_of_type[6] = {0, 0, 0, 0, 0, 0};       // dense-boundary algorithmic bytes per class, from the pictures' own maps
};

struct leon_decoder {
    leon_config cfg{};
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // display conversion overlapped with reconstruction (leon_set_overlap_convert)
    bool overlap_convert = false;
    hipStream_t conv_stream = nullptr;
    hipEvent_t ev_recon_done = nullptr;   // recorded on `stream`, waited on by conv_stream
    hipEvent_t ev_conv_done = nullptr;    // recorded on conv_stream after each conversion
    std::vector<uint8_t> conv_pending;    // per slot: a conversion may still be reading it
    Geom geom{};
    size_t plane_bytes = 0;      // cw*ch*3/2
    size_t slot_stride = 0;      // padded
    uint8_t* d_slots = nullptr;
    void* d_slots_alloc = nullptr;     // what big_alloc returned (d_slots may start inside it: LEON_SLOT_ALIGN / LEON_SLOT_SKEW)
    std::vector<uint8_t> inuse;
    // batch independence check (check_batch): which picture of the current batch writes a slot
    std::vector<uint32_t> writer_epoch;
    std::vector<int32_t> writer_of;
    uint32_t epoch = 0;
    Tables h_tables{};
    Tables* d_tables = nullptr;       // matrix set 0 (leon_set_quant_matrices) + the conversion tables
    uint8_t qm[128];
    // matrix sets 1 .. kMaxQSets - 1 (leon_add_quant_matrices): the sequences of a stream whose headers carry other
    // matrices than the first (decoders/jsv.js:540-558); a picture names its set (leon_picture.qm_set)
    static constexpr int kMaxQSets = 256;
    QTables* d_qsets = nullptr;       // [kMaxQSets], entry 0 unused
    QTables* h_qsets = nullptr;       // pinned mirror: the source of the asynchronous uploads stays valid
    std::vector<std::array<uint8_t, 128>> qsets;      // [0] unused; the matrices of set i, for the lookup
    // host-submit staging ring
    static constexpr int kStages = 4;
    Staging stages[kStages];
    size_t stage_bytes = 0;
    int next_stage = 0;
    PicDesc* d_desc_ring = nullptr;   // kDescRing descriptors for ad-hoc submits
    PicDesc* h_desc_pinned = nullptr;
    // A lap of the ring is 65536 pictures; a range is reused only after the launches that read it have finished -- not by
    // waiting for the STREAM at the wrap (the pipeline lost five windows' time, 45 ms, every 65536 pictures that way) but
    // for the event recorded behind the last launch of that range, a lap ago: long since signalled.
    static constexpr int kDescRing = 65536;
    static constexpr int kDescEventEvery = 2048;       // descriptors per event
    struct DescUse { int begin, end; hipEvent_t ev; };
    std::deque<DescUse> desc_uses;                      // in the order they were recorded
    std::vector<hipEvent_t> desc_ev_pool;
    int desc_head = 0, desc_open_begin = 0, desc_open_end = 0;      // [open_begin, open_end): committed, no event behind it yet
    // rgba
    int32_t* d_slot_ids = nullptr;
    int32_t* h_slot_ids = nullptr;
    static constexpr int kSlotIdRing = 8192;
    int slot_id_head = 0;
    uint8_t* d_rgba_tmp = nullptr;
    // timing
    bool timing = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

// nullptr (and the thread's error text) when the runtime cannot give out another event
hipEvent_t get_event(leon_decoder* d)
{
    if (!d->ev_pool.empty()) {
        hipEvent_t e = d->ev_pool.back();
        d->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
        fail(LEON_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(rc));
        return nullptr;
    }
    return e;
}

// both events of a timed launch, or LEON_ERR_HIP with nothing leaked
int get_event_pair(leon_decoder* d, hipEvent_t& a, hipEvent_t& b)
{
    a = get_event(d);
    if (!a) return LEON_ERR_HIP;
    b = get_event(d);
    if (!b) {
        d->ev_pool.push_back(a);
        a = nullptr;
        return LEON_ERR_HIP;
    }
    return LEON_OK;
}

void fill_qtables(QTables& q, const uint8_t* qm128)
{
    memset(&q, 0, sizeof q);
    for (int c = 0; c < 8; c++)
        for (int i = 0; i < 8; i++) {
            q.qmT[0][c][i] = qm128[i * 8 + c];
            q.qmT[1][c][i] = qm128[64 + i * 8 + c];
            q.pmT[c][i] = kPremultiplier[i * 8 + c];
        }
}

void upload_tables(leon_decoder* d)
{
    fill_qtables(d->h_tables.q, d->qm);
    static_assert(LEON_RGBA_LUT_SHIFT == kLutShift && sizeof(kLeonRgbaLut) == sizeof(d->h_tables.rgba_lut), "leon_rgba_lut.h and leon_kernels.h disagree");
    memcpy(d->h_tables.rgba_lut, kLeonRgbaLut, sizeof(kLeonRgbaLut));
}

inline bool alpha_geom(const leon_decoder* d) { return d->geom.alpha != 0; }
// Occupancy of the fused display kernels, set through extra dynamic LDS per workgroup (0 = what registers and LDS allow: 8 workgroups
// of 4 waves per CU, 7 for B).  Round 4, one box, per type (tools/ab_run.py tree@LEON_LDS_PAD_I=...): the I and the P launches are
// FASTER with fewer waves in flight -- I 0.477 ms at 8 workgroups per CU, 0.465 at 6, 0.458 at 5 and at 4, 0.630 at 3; P 0.581 /
// 0.572 / 0.557 / 0.557 / 0.737 -- they wait for the memory pipe, not for instructions, and 32 waves per CU streaming through ~12 regions
// each get in each other's way in L2 (dense boundary; the SPARSE kernels keep their occupancy: in the pipeline they share the CUs with
// the parser's 35 KB workgroups, and padded they cost it 8-10 % end to end); the B launches (bound by their instruction count) are indifferent between 7 and 6 (1.045 / 1.038
// ms) and lose 7 % at 5.  So: I and P at 5 workgroups per CU (20 KB + 11.5 KB of LDS per workgroup), B as the registers allow.
// Measured again at the end of round 4, when a workgroup no longer began with a round trip to memory for its tables:
// the I launch 0.382-0.384 ms at 5 workgroups per CU, **0.352 at 6**, 0.359 at 7
// -- 6 it is (20 KB + 6 KB).  P (the two-tile kernel: launch_recon_type's own pad) is the same at 5 and 6.
constexpr size_t kOccupancyPadI = 6144, kOccupancyPadP = 11776, kOccupancyPadB = 0;

int n_groups_of(const Geom& G) { return 2 * G.tasksY + 2 * G.tasksC + (G.alpha ? 2 * G.tasksY : 0); }

// Algorithmic bytes of one picture's reconstruction (dense boundary) from its own maps.  Maps in
// device memory are read back (synchronously: only prepared batches and timed ad-hoc submits ask).
int algo_bytes_of(leon_decoder* d, const leon_picture& p, bool device_maps, double& bytes)
{
    // yuva: the A plane is a second luma -- 4 of a macroblock's 10 blocks -- priced like the 4 luma blocks:
    // (512 coefficients + 256 written [+ 256 per reference]) on top of the 6-block figures below
    const size_t mbs = (size_t)d->geom.mbw * d->geom.mbh;
    // fused display conversion: + 1024 B of RGBA per macroblock, - the 384 B of planes when they are not written
    const double disp = p.rgba_out ? (1024.0 - (p.no_planes ? 384.0 : 0.0)) * (double)mbs : 0.0;
    const double a_base = d->geom.alpha ? 768.0 * (double)mbs : 0.0, a_ref = d->geom.alpha ? 256.0 : 0.0;
    if (p.type == LEON_PIC_I) {
        bytes = kBytesI * (double)mbs + disp + a_base;
        return LEON_OK;
    }
    std::vector<uint8_t> rep(mbs), dir;
    const uint8_t *rp = p.repadd, *dp = p.mb_dir;
    if (device_maps) {
        HIP_TRY(hipMemcpy(rep.data(), p.repadd, mbs, hipMemcpyDefault));
        rp = rep.data();
        if (p.type == LEON_PIC_B) {
            dir.resize(mbs);
            HIP_TRY(hipMemcpy(dir.data(), p.mb_dir, mbs, hipMemcpyDefault));
            dp = dir.data();
        }
    }
    uint64_t refs = 0;
    for (size_t i = 0; i < mbs; i++) {
        if (rp[i] >= 128) continue;                                    // replace: no prediction
        refs += p.type == LEON_PIC_P ? 1u : (uint64_t)((dp[i] & 1) + ((dp[i] >> 1) & 1));
    }
    bytes = (p.type == LEON_PIC_P ? kBytesPBase : kBytesBBase) * (double)mbs + (kBytesRef + a_ref) * (double)refs + disp + a_base;
    return LEON_OK;
}

int check_pic(const leon_decoder* d, const AnyPic& a)
{
    const leon_picture& p = a.p;
    if (p.type < LEON_PIC_I || p.type > LEON_PIC_B) return fail(LEON_ERR_INVALID, "picture type %d", p.type);
    const bool writes_planes = !(p.rgba_out && p.no_planes);
    if (writes_planes ? (p.out_slot < 0 || p.out_slot >= d->cfg.n_slots) : (p.out_slot < -1 || p.out_slot >= d->cfg.n_slots))
        return fail(LEON_ERR_INVALID, "out_slot %d", p.out_slot);
    if (a.sparse) {
        if (!a.grp_off || (!a.entries && a.n_entries) || !p.qscale || !p.intra) return fail(LEON_ERR_INVALID, "null boundary tensor");
        if (a.n_entries > (uint32_t)d->geom.cw * (uint32_t)d->geom.ch * (d->geom.alpha ? 5u : 3u) / 2u)
            return fail(LEON_ERR_INVALID, "%u entries exceed the coefficient count of a picture", a.n_entries);
    } else if (!p.coef_y || !p.coef_cb || !p.coef_cr || !p.qscale || !p.intra) return fail(LEON_ERR_INVALID, "null boundary tensor");
    else if (d->geom.alpha && !p.coef_a) return fail(LEON_ERR_INVALID, "a yuva decoder needs coef_a");
    if (p.type != LEON_PIC_I) {
        if (p.ref_fwd_slot < 0 || p.ref_fwd_slot >= d->cfg.n_slots) return fail(LEON_ERR_INVALID, "ref_fwd_slot %d", p.ref_fwd_slot);
        if (!p.repadd || !p.mv_fwd) return fail(LEON_ERR_INVALID, "P/B picture without repadd/mv_fwd");
        if (writes_planes && p.ref_fwd_slot == p.out_slot) return fail(LEON_ERR_INVALID, "out_slot equals ref_fwd_slot");
    }
    if (p.type == LEON_PIC_B) {
        if (p.ref_bwd_slot < 0 || p.ref_bwd_slot >= d->cfg.n_slots) return fail(LEON_ERR_INVALID, "ref_bwd_slot %d", p.ref_bwd_slot);
        if (!p.mv_bwd || !p.mb_dir) return fail(LEON_ERR_INVALID, "B picture without mv_bwd/mb_dir");
        if (writes_planes && p.ref_bwd_slot == p.out_slot) return fail(LEON_ERR_INVALID, "out_slot equals ref_bwd_slot");
    }
    if (p.qm_set < 0 || p.qm_set > (int32_t)d->qsets.size() - 1 + (d->qsets.empty() ? 1 : 0))
        return fail(LEON_ERR_INVALID, "qm_set %d: the decoder has %zu matrix sets beside set 0 (leon_add_quant_matrices)", p.qm_set, d->qsets.empty() ? (size_t)0 : d->qsets.size() - 1);
    if (p.rgba_out) {
        if (d->cfg.frame_width & 7) return fail(LEON_ERR_INVALID, "fused display conversion needs frame_width %% 8 == 0 (it is %d)", d->cfg.frame_width);
        if ((size_t)p.rgba_out & 15) return fail(LEON_ERR_INVALID, "rgba_out must be 16-byte aligned");
    } else if (p.no_planes) return fail(LEON_ERR_INVALID, "no_planes without rgba_out: the picture would leave nothing behind");
    return LEON_OK;
}

void fill_desc(const leon_decoder* d, const AnyPic& a, PicDesc& o)
{
    const leon_picture& p = a.p;
    o.grp_off = a.grp_off;
    o.entries = a.entries;
    o.n_entries = a.n_entries;
    o.coef[0] = p.coef_y;
    o.coef[1] = p.coef_cb;
    o.coef[2] = p.coef_cr;
    o.coef[3] = p.coef_a;
    o.qscale = p.qscale;
    o.intra = p.intra;
    o.repadd = p.repadd;
    o.mb_dir = p.mb_dir;
    o.mv_fwd = p.mv_fwd;
    o.mv_bwd = p.mv_bwd;
    o.out = d->d_slots + (size_t)(p.out_slot < 0 ? 0 : p.out_slot) * d->slot_stride;   // not written when no_planes
    o.ref_fwd = p.type != LEON_PIC_I ? d->d_slots + (size_t)p.ref_fwd_slot * d->slot_stride : nullptr;
    o.ref_bwd = p.type == LEON_PIC_B ? d->d_slots + (size_t)p.ref_bwd_slot * d->slot_stride : nullptr;
    o.type = p.type;
    o.rgba = (uint8_t*)p.rgba_out;
    o.no_planes = p.rgba_out ? p.no_planes : 0;
    o.pad_ = 0;
    o.qt = p.qm_set > 0 ? d->d_qsets + p.qm_set : &d->d_tables->q;
}

// a submit that overwrites a slot still being converted on the second stream waits for it
int guard_pending_conversions(leon_decoder* d, const int32_t* out_slots, int n)
{
    if (!d->overlap_convert || d->conv_pending.empty()) return LEON_OK;
    bool hit = false;
    for (int i = 0; i < n && !hit; i++) hit = out_slots[i] >= 0 && d->conv_pending[out_slots[i]] != 0;
    if (hit) {
        HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_conv_done, 0));
        std::fill(d->conv_pending.begin(), d->conv_pending.end(), 0);
    }
    return LEON_OK;
}

// one launch of the type-specialised kernel over n pictures of that type
int launch_recon_type(leon_decoder* d, int type, const PicDesc* d_descs, int n, double dense_bytes, bool sparse = false, uint64_t entries = 0,
                      bool display = false)
{
    Geom G = d->geom;
    G.n_pics = n;
    const bool alpha = d->geom.alpha != 0;
    const bool pair = LEON_PAIR_LUMA && LEON_CARRY && display && !sparse && !alpha && type != LEON_PIC_I;      // k_recon_display: recon_luma_pair, two tiles
    // waves per workgroup: 4 -- but 6 for the dense B display kernel, whose two tiles per wave would leave room for 5 workgroups of 4
    // per CU (20 waves: it loses 7 % there) and leave room for 4 of 6 (24 waves, what its registers allow anyway); LEON_WAVES_B (experiments)
    static const int waves_b = getenv("LEON_WAVES_B") ? atoi(getenv("LEON_WAVES_B")) : kWavesPerWGPairB;      // (6 measured: +5 % on the B launches; 4 it is)
    const int wpw = pair && type == LEON_PIC_B && waves_b >= 1 && waves_b * 64 <= kReconMaxThreads ? waves_b : kWavesPerWG;
    if (display) {   // one task = one chroma group with everything above it (k_recon_display)
        G.tasks_per_pic = G.tasksC;
        G.wg_per_pic = (G.tasks_per_pic + wpw - 1) / wpw;
        G.inv_wg_per_pic = G.wg_per_pic == 1 ? 0u : (uint32_t)(((1ull << 32) + G.wg_per_pic - 1) / G.wg_per_pic);
    }
    // B launches: the workgroups of two consecutive pictures alternate (pic_of_wg), an odd last picture leaves its partner's idle
    long long wgs = (long long)(type == LEON_PIC_B && LEON_PAIR_B ? (n + 1) / 2 * 2 : n) * G.wg_per_pic;
    // the kernel divides by multiply-high: exact while n_wg * wg_per_pic < 2^32
    if (wgs > 0x7fffffffLL || wgs * G.wg_per_pic >= (1LL << 32))
        return fail(LEON_ERR_INVALID, "batch of %d pictures is too large for one launch; split it", n);
    G.n_wg = (int)wgs;
    TimedLaunch tl{};
    if (d->timing) {
        if (get_event_pair(d, tl.a, tl.b) != LEON_OK) return LEON_ERR_HIP;
        tl.kind = 0;
        tl.pic_type = type;
        tl.mbs = (uint64_t)d->geom.mbw * d->geom.mbh * (uint64_t)n;
        tl.bytes = dense_bytes;
        if (sparse)   // the lists replace the 768 B/MB of dense coefficients
            tl.bytes += 4.0 * (double)entries + 4.0 * (double)(n_groups_of(d->geom) + 1) * n - 768.0 * (double)tl.mbs;
        HIP_TRY(hipEventRecord(tl.a, d->stream));
    }
    static_assert(64 * kWavesPerWG <= kReconMaxThreads && 64 * kWavesPerWGPairB <= kReconMaxThreads, "k_recon is launched with more threads than its __launch_bounds__");
    const dim3 grid(G.n_wg), block(64 * wpw);
    // LEON_DEBUG_LDS_PAD (bytes): extra dynamic LDS per workgroup = an occupancy throttle for experiments
    static const size_t lds_pad_all = getenv("LEON_DEBUG_LDS_PAD") ? (size_t)atol(getenv("LEON_DEBUG_LDS_PAD")) : 0;
    // per picture type (experiments): LEON_LDS_PAD_I / _P / _B
    static const size_t lds_pad_t[3] = {getenv("LEON_LDS_PAD_I") ? (size_t)atol(getenv("LEON_LDS_PAD_I")) : kOccupancyPadI,
                                        getenv("LEON_LDS_PAD_P") ? (size_t)atol(getenv("LEON_LDS_PAD_P")) : kOccupancyPadP,
                                        getenv("LEON_LDS_PAD_B") ? (size_t)atol(getenv("LEON_LDS_PAD_B")) : kOccupancyPadB};
    const size_t lds_pad = lds_pad_all + (display && !sparse && !alpha_geom(d) ? lds_pad_t[type - 1] : 0);
    // the two-tile kernels (26.25 KB per workgroup = six per CU): P back to five (LEON_LDS_PAD_PP / _BP: experiments)
    static const size_t lds_pad_ppair = getenv("LEON_LDS_PAD_PP") ? (size_t)atol(getenv("LEON_LDS_PAD_PP")) : 1024;
    static const size_t lds_pad_bpair = getenv("LEON_LDS_PAD_BP") ? (size_t)atol(getenv("LEON_LDS_PAD_BP")) : 0;
    const size_t lds = (size_t)wpw * (display ? (alpha ? kLdsPerWaveDisplayAlpha : (pair ? kLdsPerWaveDisplayPair : kLdsPerWaveDisplay)) : kLdsPerWave) + (pair ? lds_pad_all + (type == LEON_PIC_P ? lds_pad_ppair : lds_pad_bpair) : lds_pad);      // display kernels: + kLdsLut of static LDS (the conversion tables)
    if (display && alpha) {          // yuva: the A parts ride in the same task (k_recon_display<.., .., true>)
        if (!sparse) {
            if (type == LEON_PIC_I) hipLaunchKernelGGL((k_recon_display<1, false, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else if (type == LEON_PIC_P) hipLaunchKernelGGL((k_recon_display<2, false, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else hipLaunchKernelGGL((k_recon_display<3, false, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        } else {
            if (type == LEON_PIC_I) hipLaunchKernelGGL((k_recon_display<1, true, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else if (type == LEON_PIC_P) hipLaunchKernelGGL((k_recon_display<2, true, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else hipLaunchKernelGGL((k_recon_display<3, true, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        }
    } else if (display) {
        if (!sparse) {
            if (type == LEON_PIC_I) hipLaunchKernelGGL((k_recon_display<1, false>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else if (type == LEON_PIC_P) hipLaunchKernelGGL((k_recon_display<2, false>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else hipLaunchKernelGGL((k_recon_display<3, false>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        } else {
            if (type == LEON_PIC_I) hipLaunchKernelGGL((k_recon_display<1, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else if (type == LEON_PIC_P) hipLaunchKernelGGL((k_recon_display<2, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
            else hipLaunchKernelGGL((k_recon_display<3, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        }
    } else if (!sparse) {
        if (type == LEON_PIC_I) hipLaunchKernelGGL((k_recon<1, false>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        else if (type == LEON_PIC_P) hipLaunchKernelGGL((k_recon<2, false>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        else hipLaunchKernelGGL((k_recon<3, false>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
    } else {
        if (type == LEON_PIC_I) hipLaunchKernelGGL((k_recon<1, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        else if (type == LEON_PIC_P) hipLaunchKernelGGL((k_recon<2, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
        else hipLaunchKernelGGL((k_recon<3, true>), grid, block, lds, d->stream, d_descs, G, d->d_tables);
    }
    HIP_TRY(hipGetLastError());
    if (d->timing) {
        HIP_TRY(hipEventRecord(tl.b, d->stream));
        d->timed.push_back(tl);
    }
    return LEON_OK;
}

// launch class of a picture: its type, and whether the display conversion is fused in
inline int class_of(const leon_picture& p) { return (p.rgba_out ? 3 : 0) + p.type - 1; }

// descriptors sorted by class; one launch per class present
int launch_recon(leon_decoder* d, const PicDesc* d_descs, const int count[6], const double bytes[6], bool sparse = false, const uint64_t* entries = nullptr)
{
    // B pictures first, anchors last: what the NEXT batch will read as references (I and P planes)
    // is then the most recently written data and still sits in the 256 MB Infinity Cache / L2,
    // instead of being pushed out by B planes nobody reads again.
    int at[6];
    for (int k = 0, a = 0; k < 6; k++) { at[k] = a; a += count[k]; }
    static const int order[6] = {5, 2, 4, 1, 3, 0};
    for (int k : order) {
        if (count[k] > 0) {
            int rc = launch_recon_type(d, k % 3 + 1, d_descs + at[k], count[k], bytes[k], sparse, entries ? entries[k] : 0, k >= 3);
            if (rc != LEON_OK) return rc;
        }
    }
    return LEON_OK;
}

// fill `out` with the descriptors of pics sorted by class, and the per-class counts
void sorted_descs(const leon_decoder* d, const AnyPic* pics, int n, PicDesc* out, int count[6], uint64_t entries[6])
{
    for (int k = 0; k < 6; k++) { count[k] = 0; entries[k] = 0; }
    for (int i = 0; i < n; i++) {
        count[class_of(pics[i].p)]++;
        entries[class_of(pics[i].p)] += pics[i].n_entries;
    }
    int at[6];
    for (int k = 0, a = 0; k < 6; k++) { at[k] = a; a += count[k]; }
    for (int i = 0; i < n; i++) fill_desc(d, pics[i], out[at[class_of(pics[i].p)]++]);
}

// Large buffers: physically contiguous when the device grants it (include/leon.h leon_device_malloc) -- by default ONLY
// what a caller asks for through leon_device_malloc / leon_config.contiguous_slots.  Which of the library's own kinds of
// buffer ask for it too: LEON_CONTIGUOUS = a mask of kBig* (A/B runs).  What round 3 measured, one box, runs alternating:
// bench.py 6.05-6.10 ms per step with the caller's buffers (and the slot ring) contiguous against 5.92-6.29 without; the
// pipeline 176 k pictures/s with slot ring + RGBA ring contiguous, 169 k with nothing, 150-155 k with the parser's arenas
// contiguous as well.
//
// CONTIGUOUS MEMORY IS NEVER RETURNED TO THE DRIVER (round 4).  Round 3: with contiguous slot rings that were hipFree'd when
// their decoder went, 13 of 30 runs of the whole GPU test suite in one process ended with a B picture of a LATER, small
// pipeline wrong in whole macroblocks (the same picture every time; none of 22 runs with ordinary rings, none of 6 with
// contiguous rings that were simply never freed).  The victim's own buffers are ordinary allocations, every stream of the
// earlier decoder was idle before its ring was freed, and every byte the B kernels read is written by this library first
// (walked again in round 4: descriptors, tables, lists, maps, slots, LDS) -- what the failing runs share is not a read of
// this library but the driver getting physically contiguous pages back in mid-process.  So it does not get them back:
// a contiguous allocation becomes a SEGMENT of a process-lifetime pool, big_free() hands its range to the next request
// (first the smallest free range that fits; ranges are split at 2 MiB granules and merged with free neighbours of their
// segment), and only ordinary allocations go through hipFree.  What that costs: a process holds the high-water mark of its
// contiguous requests until it exits (leon_device_pool_stats reports it).
enum { kBigSlots = 1, kBigRgbaRing = 2, kBigArenas = 4, kBigCaller = 8 };

struct ContigPool {
    struct Range { char* ptr; size_t bytes; int dev; int seg; bool free; };
    std::mutex mu;
    std::vector<Range> ranges;      // ordered by (segment, address): neighbours in the vector are neighbours in memory
    int n_segments = 0;
    size_t held = 0, in_use = 0;    // bytes taken from the driver / handed out
    static constexpr size_t kGranule = (size_t)2 << 20;

    // a free range of the device that fits, the smallest one; nullptr: none
    void* take(size_t need, int dev)
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < ranges.size(); i++)
            if (ranges[i].free && ranges[i].dev == dev && ranges[i].bytes >= need && (best == (size_t)-1 || ranges[i].bytes < ranges[best].bytes)) best = i;
        if (best == (size_t)-1) return nullptr;
        if (ranges[best].bytes > need) {       // split: the tail stays free
            Range tail = ranges[best];
            tail.ptr += need;
            tail.bytes -= need;
            ranges[best].bytes = need;
            ranges.insert(ranges.begin() + (long)best + 1, tail);
        }
        ranges[best].free = false;
        in_use += need;
        return ranges[best].ptr;
    }
    void add_segment(void* p, size_t bytes, int dev)      // a fresh contiguous allocation, handed out whole
    {
        std::lock_guard<std::mutex> lk(mu);
        ranges.push_back(Range{(char*)p, bytes, dev, n_segments++, false});
        held += bytes;
        in_use += bytes;
    }
    int device_of(const void* p)          // of a range that is handed out, -1: not ours
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Range& r : ranges)
            if (r.ptr == (const char*)p && !r.free) return r.dev;
        return -1;
    }
    // true: p was one of ours and is free again (merged with free neighbours of its segment)
    bool give_back(void* p)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < ranges.size(); i++) {
            if (ranges[i].ptr != (char*)p || ranges[i].free) continue;
            ranges[i].free = true;
            in_use -= ranges[i].bytes;
            if (i + 1 < ranges.size() && ranges[i + 1].free && ranges[i + 1].seg == ranges[i].seg) {
                ranges[i].bytes += ranges[i + 1].bytes;
                ranges.erase(ranges.begin() + (long)i + 1);
            }
            if (i > 0 && ranges[i - 1].free && ranges[i - 1].seg == ranges[i].seg) {
                ranges[i - 1].bytes += ranges[i].bytes;
                ranges.erase(ranges.begin() + (long)i);
            }
            return true;
        }
        return false;
    }
};
ContigPool& contig_pool()
{
    static ContigPool* pool = new ContigPool();       // never destroyed: its memory lives as long as the process
    return *pool;
}

hipError_t big_alloc(void** ptr, size_t bytes, int kind, bool* contiguous = nullptr, bool asked = false)
{
    static const int mask = getenv("LEON_CONTIGUOUS") ? atoi(getenv("LEON_CONTIGUOUS")) : kBigCaller;
    // LEON_DEBUG_ZERO_ALLOC = a mask of kBig*: such buffers start as zeros (hunting reads of memory nobody wrote)
    static const int zero = getenv("LEON_DEBUG_ZERO_ALLOC") ? atoi(getenv("LEON_DEBUG_ZERO_ALLOC")) : 0;
    if (contiguous) *contiguous = false;
    hipError_t e = hipErrorOutOfMemory;
    if (((mask & kind) || asked) && bytes >= ((size_t)1 << 20)) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const size_t need = (bytes + ContigPool::kGranule - 1) / ContigPool::kGranule * ContigPool::kGranule;
        if (void* p = contig_pool().take(need, dev)) {
            *ptr = p;
            e = hipSuccess;
        } else {
            e = hipExtMallocWithFlags(ptr, need, hipDeviceMallocContiguous);
            if (e == hipSuccess) contig_pool().add_segment(*ptr, need, dev);
            else (void)hipGetLastError();
        }
        if (e == hipSuccess && contiguous) *contiguous = true;
    }
    if (e != hipSuccess) e = hipMalloc(ptr, bytes);
    if (e == hipSuccess && (zero & kind)) { (void)hipMemset(*ptr, 0, bytes); (void)hipDeviceSynchronize(); }
    return e;
}

// The counterpart of big_alloc.  The caller has made sure nothing in flight touches the buffer (every stream that used it
// is idle): a pooled range goes straight to the next request, and hipFree's own wait for the device is not leaned on.
hipError_t big_free(void* p)
{
    if (!p) return hipSuccess;
    if (contig_pool().give_back(p)) return hipSuccess;
    return hipFree(p);
}

// an event behind everything committed so far
int close_desc_range(leon_decoder* d)
{
    if (d->desc_open_end == d->desc_open_begin) return LEON_OK;
    hipEvent_t ev = nullptr;
    if (!d->desc_ev_pool.empty()) { ev = d->desc_ev_pool.back(); d->desc_ev_pool.pop_back(); }
    else HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, d->stream));
    d->desc_uses.push_back(leon_decoder::DescUse{d->desc_open_begin, d->desc_open_end, ev});
    d->desc_open_begin = d->desc_open_end;
    return LEON_OK;
}

// reserve n consecutive descriptors in the ring; what lay there a lap ago must have been read
int reserve_descs(leon_decoder* d, int n, int& at)
{
    if (n > leon_decoder::kDescRing) return fail(LEON_ERR_INVALID, "batch of %d pictures exceeds %d; use leon_batch_create", n, leon_decoder::kDescRing);
    auto retire = [&](void) -> int {
        leon_decoder::DescUse u = d->desc_uses.front();
        d->desc_uses.pop_front();
        HIP_TRY(hipEventSynchronize(u.ev));
        d->desc_ev_pool.push_back(u.ev);
        return LEON_OK;
    };
    if (d->desc_head + n > leon_decoder::kDescRing) {                 // wrap: the tail stays unused this lap
        int rc = close_desc_range(d);
        if (rc != LEON_OK) return rc;
        while (d->desc_uses.size() > 1 && d->desc_uses.front().begin > d->desc_uses.back().begin)
            if ((rc = retire()) != LEON_OK) return rc;               // ranges of the lap BEFORE this one, behind where it ended: the oldest there are
        d->desc_head = 0;
        d->desc_open_begin = d->desc_open_end = 0;
    }
    at = d->desc_head;
    if (d->desc_open_end > d->desc_open_begin && at < d->desc_open_end && at + n > d->desc_open_begin) {
        const int rc = close_desc_range(d);
        if (rc != LEON_OK) return rc;
    }
    while (!d->desc_uses.empty() && d->desc_uses.front().begin < at + n && d->desc_uses.front().end > at) {
        const int rc = retire();
        if (rc != LEON_OK) return rc;
    }
    d->desc_head += n;
    return LEON_OK;
}

// the launches that read [at, at + n) have been queued
int commit_descs(leon_decoder* d, int at, int n)
{
    if (d->desc_open_end == d->desc_open_begin) d->desc_open_begin = at;
    d->desc_open_end = at + n;
    return d->desc_open_end - d->desc_open_begin >= leon_decoder::kDescEventEvery ? close_desc_range(d) : LEON_OK;
}

}  // namespace

extern "C" {

int leon_abi_version(void) { return LEON_ABI_VERSION; }

const char* leon_last_error(void) { return g_err.c_str(); }

int leon_create(const leon_config* cfg, leon_decoder** out)
{
    if (!cfg || !out) return fail(LEON_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->coded_width <= 0 || cfg->coded_height <= 0 || (cfg->coded_width & 15) || (cfg->coded_height & 15))
        return fail(LEON_ERR_INVALID, "coded size %dx%d must be positive multiples of 16", cfg->coded_width, cfg->coded_height);
    // the format's size fields are 12 bits (decoders/jsv.js:491-500): coded sizes end at 4096
    if (cfg->coded_width > 4096 || cfg->coded_height > 4096)
        return fail(LEON_ERR_INVALID, "coded size %dx%d exceeds the format's 4096x4096", cfg->coded_width, cfg->coded_height);
    if (cfg->frame_width <= 0 || cfg->frame_height <= 0 || cfg->frame_width > cfg->coded_width || cfg->frame_height > cfg->coded_height)
        return fail(LEON_ERR_INVALID, "frame size %dx%d", cfg->frame_width, cfg->frame_height);
    if (cfg->n_slots < 1) return fail(LEON_ERR_INVALID, "n_slots %d", cfg->n_slots);
    if (cfg->alpha && (cfg->frame_width & 1)) return fail(LEON_ERR_INVALID, "a yuva decoder needs an even frame width (it is %d)", cfg->frame_width);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(LEON_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(LEON_ERR_NO_DEVICE, "device %d of %d", cfg->device_id, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return fail(LEON_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LEON_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", cfg->device_id, prop.gcnArchName);
    HIP_TRY(hipSetDevice(cfg->device_id));

    leon_decoder* d = new (std::nothrow) leon_decoder();
    if (!d) return fail(LEON_ERR_NOMEM, "out of host memory");
    d->cfg = *cfg;
    d->dev = cfg->device_id;
    if (cfg->stream) {
        d->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
            delete d;
            return fail(LEON_ERR_HIP, "hipStreamCreate failed");
        }
        d->own_stream = true;
    }
    Geom& G = d->geom;
    G.cw = cfg->coded_width;
    G.ch = cfg->coded_height;
    G.mbw = G.cw >> 4;
    G.mbh = G.ch >> 4;
    G.gY = ((G.cw >> 3) + 7) >> 3;
    G.gC = ((G.cw >> 4) + 7) >> 3;
    G.tasksY = G.gY * G.mbh;            // one task = both block rows of a macroblock row
    G.tasksC = G.gC * G.mbh;            // one task = the Cb and the Cr group of a block row
    G.alpha = cfg->alpha ? 1 : 0;
    G.tasks_per_pic = G.tasksY + G.tasksC + (G.alpha ? G.tasksY : 0);   // yuva: the A plane's luma-shaped tasks last
    G.wg_per_pic = (G.tasks_per_pic + kWavesPerWG - 1) / kWavesPerWG;
    auto inv32 = [](uint32_t dv) { return (uint32_t)(((1ull << 32) + dv - 1) / dv); };   // exact for n*dv < 2^32
    G.inv_wg_per_pic = G.wg_per_pic == 1 ? 0u : inv32((uint32_t)G.wg_per_pic);
    G.inv_gY = G.gY == 1 ? 0u : inv32((uint32_t)G.gY);
    G.inv_gC = G.gC == 1 ? 0u : inv32((uint32_t)G.gC);
    G.fw = cfg->frame_width;
    G.fh = cfg->frame_height;
    d->plane_bytes = (size_t)G.cw * G.ch * (G.alpha ? 5 : 3) / 2;
    d->slot_stride = (d->plane_bytes + 255) / 256 * 256 + 256;   // tail pad: the 12-byte MC window may over-read 3 bytes
    d->inuse.assign(cfg->n_slots, 0);
    auto bail = [&](const char* what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(hipGetLastError());
        leon_destroy(d);
        return fail(LEON_ERR_NOMEM, "%s", msg.c_str());
    };
    // LEON_SLOT_ALIGN / LEON_SLOT_SKEW (bytes; placement experiments, tools/probe/placement_probe.py): the ring starts at a multiple
    // of ALIGN plus SKEW inside a larger allocation
    const size_t s_align = getenv("LEON_SLOT_ALIGN") ? (size_t)atoll(getenv("LEON_SLOT_ALIGN")) : 0, s_skew = getenv("LEON_SLOT_SKEW") ? (size_t)atoll(getenv("LEON_SLOT_SKEW")) : 0;
    if (big_alloc((void**)&d->d_slots_alloc, d->slot_stride * (size_t)cfg->n_slots + 256 + s_align + s_skew, kBigSlots, nullptr, cfg->contiguous_slots == 1) != hipSuccess) return bail("slot ring");
    d->d_slots = (uint8_t*)d->d_slots_alloc;
    if (s_align) d->d_slots = (uint8_t*)(((uintptr_t)d->d_slots + s_align - 1) / s_align * s_align);
    d->d_slots += s_skew;
    if (hipMemsetAsync(d->d_slots, 0, d->slot_stride * (size_t)cfg->n_slots + 256, d->stream) != hipSuccess) return bail("slot memset");
    if (hipMalloc(&d->d_tables, sizeof(Tables)) != hipSuccess) return bail("tables");
    memcpy(d->qm, kDefaultIntra, 64);
    memset(d->qm + 64, 16, 64);
    upload_tables(d);
    if (hipMemcpyAsync(d->d_tables, &d->h_tables, sizeof(Tables), hipMemcpyHostToDevice, d->stream) != hipSuccess) return bail("tables upload");
    if (hipMalloc(&d->d_desc_ring, sizeof(PicDesc) * leon_decoder::kDescRing) != hipSuccess) return bail("descriptor ring");
    if (hipHostMalloc((void**)&d->h_desc_pinned, sizeof(PicDesc) * leon_decoder::kDescRing) != hipSuccess) return bail("pinned descriptors");
    if (hipMalloc(&d->d_slot_ids, sizeof(int32_t) * leon_decoder::kSlotIdRing) != hipSuccess) return bail("slot id ring");
    if (hipHostMalloc((void**)&d->h_slot_ids, sizeof(int32_t) * leon_decoder::kSlotIdRing) != hipSuccess) return bail("pinned slot ids");
    // staging for host-memory pictures: coef planes (2 bytes/sample) + 5 byte maps + 2 vector maps
    size_t mbs = (size_t)G.mbw * G.mbh;
    // coefficients: dense planes (2 B each) or, at worst, one 4-byte entry each plus the group offsets
    d->stage_bytes = d->plane_bytes * 4 + (G.alpha ? (size_t)G.cw * G.ch * 2 : 0) + ((size_t)n_groups_of(d->geom) + 1) * 4 + 1024 + 4 * ((mbs + 255) / 256 * 256) + 2 * ((mbs * 4 + 255) / 256 * 256) + 1024;
    if (hipStreamSynchronize(d->stream) != hipSuccess) return bail("create sync");
    *out = d;
    return LEON_OK;
}

void leon_destroy(leon_decoder* d)
{
    if (!d) return;
    hipSetDevice(d->dev);
    // nothing of this decoder may still run when its memory goes: both streams, explicitly (hipFree's own wait for the
    // device is not something to lean on -- and not promised for memory from hipExtMallocWithFlags)
    if (d->stream) hipStreamSynchronize(d->stream);
    if (d->conv_stream) hipStreamSynchronize(d->conv_stream);
    for (auto& t : d->timed) {
        hipEventDestroy(t.a);
        hipEventDestroy(t.b);
    }
    for (auto e : d->ev_pool) hipEventDestroy(e);
    for (auto& u : d->desc_uses) hipEventDestroy(u.ev);
    for (auto e : d->desc_ev_pool) hipEventDestroy(e);
    for (auto& s : d->stages) {
        if (s.base) hipFree(s.base);
        if (s.host) hipHostFree(s.host);
        if (s.done) hipEventDestroy(s.done);
    }
    if (d->d_slots_alloc) big_free(d->d_slots_alloc);          // a contiguous ring goes back to the process's pool, never to the driver (big_alloc)
    if (d->d_tables) hipFree(d->d_tables);
    if (d->d_qsets) hipFree(d->d_qsets);
    if (d->h_qsets) hipHostFree(d->h_qsets);
    if (d->d_desc_ring) hipFree(d->d_desc_ring);
    if (d->h_desc_pinned) hipHostFree(d->h_desc_pinned);
    if (d->d_slot_ids) hipFree(d->d_slot_ids);
    if (d->h_slot_ids) hipHostFree(d->h_slot_ids);
    if (d->d_rgba_tmp) hipFree(d->d_rgba_tmp);
    if (d->conv_stream) {
        hipStreamDestroy(d->conv_stream);
        hipEventDestroy(d->ev_recon_done);
        hipEventDestroy(d->ev_conv_done);
    }
    if (d->own_stream && d->stream) hipStreamDestroy(d->stream);
    delete d;
}

int leon_set_quant_matrices(leon_decoder* d, const uint8_t* intra64, const uint8_t* non_intra64)
{
    if (!d) return fail(LEON_ERR_INVALID, "null decoder");
    HIP_TRY(hipSetDevice(d->dev));
    if (intra64) memcpy(d->qm, intra64, 64);
    if (non_intra64) memcpy(d->qm + 64, non_intra64, 64);
    // in-flight kernels read d_tables: order the update behind them, and keep the
    // host copy stable until the copy has been consumed
    HIP_TRY(hipStreamSynchronize(d->stream));
    upload_tables(d);
    HIP_TRY(hipMemcpyAsync(d->d_tables, &d->h_tables, sizeof(Tables), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LEON_OK;
}

int leon_add_quant_matrices(leon_decoder* d, const uint8_t* intra64, const uint8_t* non_intra64, int32_t* set)
{
    if (!d || !set) return fail(LEON_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(d->dev));
    std::array<uint8_t, 128> m;
    memcpy(m.data(), intra64 ? intra64 : kDefaultIntra, 64);
    if (non_intra64) memcpy(m.data() + 64, non_intra64, 64); else memset(m.data() + 64, 16, 64);
    for (size_t i = 1; i < d->qsets.size(); i++)
        if (d->qsets[i] == m) { *set = (int32_t)i; return LEON_OK; }
    if (!d->d_qsets) {
        HIP_TRY(hipMalloc(&d->d_qsets, sizeof(QTables) * leon_decoder::kMaxQSets));
        HIP_TRY(hipHostMalloc((void**)&d->h_qsets, sizeof(QTables) * leon_decoder::kMaxQSets));
        d->qsets.assign(1, std::array<uint8_t, 128>{});
    }
    if ((int)d->qsets.size() >= leon_decoder::kMaxQSets)
        return fail(LEON_ERR_INVALID, "a decoder holds at most %d different sets of quantiser matrices", leon_decoder::kMaxQSets - 1);
    const size_t id = d->qsets.size();
    fill_qtables(d->h_qsets[id], m.data());
    // a set is written once and never changed: the upload is ordered in front of the launches that name it by the stream
    HIP_TRY(hipMemcpyAsync(d->d_qsets + id, d->h_qsets + id, sizeof(QTables), hipMemcpyHostToDevice, d->stream));
    d->qsets.push_back(m);
    *set = (int32_t)id;
    return LEON_OK;
}

int leon_acquire_slot(leon_decoder* d, int32_t* slot)
{
    if (!d || !slot) return fail(LEON_ERR_INVALID, "null argument");
    for (int j = 0; j < d->cfg.n_slots; j++)
        if (!d->inuse[j]) {
            d->inuse[j] = 1;
            *slot = j;
            return LEON_OK;
        }
    return fail(LEON_ERR_NO_FREE_SLOT, "no free render buffers");
}

int leon_release_slot(leon_decoder* d, int32_t slot)
{
    if (!d || slot < 0 || slot >= d->cfg.n_slots) return fail(LEON_ERR_INVALID, "slot %d", slot);
    d->inuse[slot] = 0;
    return LEON_OK;
}

int leon_free_decoded_slots(leon_decoder* d)
{
    if (!d) return fail(LEON_ERR_INVALID, "null decoder");
    std::fill(d->inuse.begin(), d->inuse.end(), 0);
    return LEON_OK;
}

}  // extern "C"

namespace {

// = IDCT_GL for one picture whose arrays are in host memory: stage, then reconstruct
int submit_picture_any(leon_decoder* d, const AnyPic& pic)
{
    HIP_TRY(hipSetDevice(d->dev));
    int rc = check_pic(d, pic);
    if (rc != LEON_OK) return rc;
    const Geom& G = d->geom;
    const size_t n_groups = (size_t)n_groups_of(G);
    if (pic.sparse) {
        // host lists can be checked: offsets ascending and closed by the list length
        uint32_t prev = 0;
        for (size_t g = 0; g <= n_groups; g++) {
            if (pic.grp_off[g] < prev) return fail(LEON_ERR_INVALID, "grp_off is not ascending at group %zu", g);
            prev = pic.grp_off[g];
        }
        if (prev != pic.n_entries) return fail(LEON_ERR_INVALID, "grp_off ends at %u, n_entries is %u", prev, pic.n_entries);
    }
    Staging& s = d->stages[d->next_stage];
    d->next_stage = (d->next_stage + 1) % leon_decoder::kStages;
    if (!s.base) {
        HIP_TRY(hipMalloc(&s.base, d->stage_bytes));
        HIP_TRY(hipHostMalloc((void**)&s.host, d->stage_bytes, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (s.busy) HIP_TRY(hipEventSynchronize(s.done));
    size_t ny = (size_t)G.cw * G.ch, nc = ny >> 2, mbs = (size_t)G.mbw * G.mbh;
    size_t mpad = (mbs + 255) / 256 * 256, vpad = (mbs * 4 + 255) / 256 * 256;
    char* p = s.base;
    AnyPic dp = pic;
    // The pieces are gathered into the pinned mirror with plain memcpy and cross PCIe in one
    // asynchronous copy: the caller gets its arrays (and its thread) back after the gather, where
    // nine copies from pageable memory would each block until staged by the runtime.
    auto put = [&](const void* src, size_t bytes, size_t reserve) -> const void* {
        char* at = p;
        p += reserve;
        if (src && bytes) memcpy(s.host + (at - s.base), src, bytes);
        return src ? at : nullptr;
    };
    auto pad256 = [](size_t v) { return (v + 255) / 256 * 256; };
    if (pic.sparse) {
        dp.grp_off = (const uint32_t*)put(pic.grp_off, (n_groups + 1) * 4, pad256((n_groups + 1) * 4));
        dp.entries = (const uint32_t*)put(pic.entries ? (const void*)pic.entries : (const void*)pic.grp_off,
                                          (size_t)pic.n_entries * 4, pad256((size_t)pic.n_entries * 4 + 4));
    } else {
        dp.p.coef_y = (const int16_t*)put(pic.p.coef_y, ny * 2, ny * 2);
        dp.p.coef_cb = (const int16_t*)put(pic.p.coef_cb, nc * 2, nc * 2);
        dp.p.coef_cr = (const int16_t*)put(pic.p.coef_cr, nc * 2, nc * 2);
        dp.p.coef_a = (const int16_t*)put(G.alpha ? pic.p.coef_a : nullptr, ny * 2, G.alpha ? ny * 2 : 0);
    }
    const int type = pic.p.type;
    dp.p.qscale = (const uint8_t*)put(pic.p.qscale, mbs, mpad);
    dp.p.intra = (const uint8_t*)put(pic.p.intra, mbs, mpad);
    dp.p.repadd = (const uint8_t*)put(type != LEON_PIC_I ? pic.p.repadd : nullptr, mbs, mpad);
    dp.p.mb_dir = (const uint8_t*)put(type == LEON_PIC_B ? pic.p.mb_dir : nullptr, mbs, mpad);
    dp.p.mv_fwd = (const int16_t*)put(type != LEON_PIC_I ? pic.p.mv_fwd : nullptr, mbs * 4, vpad);
    dp.p.mv_bwd = (const int16_t*)put(type == LEON_PIC_B ? pic.p.mv_bwd : nullptr, mbs * 4, vpad);
    HIP_TRY(hipMemcpyAsync(s.base, s.host, (size_t)(p - s.base), hipMemcpyHostToDevice, d->stream));
    int at = 0;
    rc = reserve_descs(d, 1, at);
    if (rc != LEON_OK) return rc;
    fill_desc(d, dp, d->h_desc_pinned[at]);
    rc = guard_pending_conversions(d, &dp.p.out_slot, 1);
    if (rc != LEON_OK) return rc;
    HIP_TRY(hipMemcpyAsync(d->d_desc_ring + at, d->h_desc_pinned + at, sizeof(PicDesc), hipMemcpyHostToDevice, d->stream));
    double bytes = 0;
    if (d->timing) {
        rc = algo_bytes_of(d, pic.p, false, bytes);
        if (rc != LEON_OK) return rc;
    }
    rc = launch_recon_type(d, type, d->d_desc_ring + at, 1, bytes, pic.sparse, pic.n_entries, pic.p.rgba_out != nullptr);
    if (rc != LEON_OK) return rc;
    rc = commit_descs(d, at, 1);
    if (rc != LEON_OK) return rc;
    HIP_TRY(hipEventRecord(s.done, d->stream));
    s.busy = true;
    return LEON_OK;
}

// The pictures of one batch run in the same launches and must be mutually independent: no two write
// the same slot, none reads a slot another one writes.  O(n) for any n: the slots written by the batch
// are stamped in a per-decoder table (stamp = picture index + 1 under a fresh epoch), then every
// reference is looked up in it.
int check_batch(leon_decoder* d, const AnyPic* pics, int n)
{
    for (int i = 0; i < n; i++) {
        int rc = check_pic(d, pics[i]);
        if (rc != LEON_OK) return rc;
    }
    if (d->writer_epoch.size() != (size_t)d->cfg.n_slots) {
        d->writer_epoch.assign(d->cfg.n_slots, 0);
        d->writer_of.assign(d->cfg.n_slots, 0);
        d->epoch = 0;
    }
    if (++d->epoch == 0) {                       // wrapped after 2^32 batches: start over
        std::fill(d->writer_epoch.begin(), d->writer_epoch.end(), 0u);
        d->epoch = 1;
    }
    for (int i = 0; i < n; i++) {
        if (pics[i].p.rgba_out && pics[i].p.no_planes) continue;       // writes no slot
        const int s = pics[i].p.out_slot;
        if (d->writer_epoch[s] == d->epoch)
            return fail(LEON_ERR_INVALID, "pictures %d and %d of one batch depend on each other (both write slot %d)", d->writer_of[s], i, s);
        d->writer_epoch[s] = d->epoch;
        d->writer_of[s] = i;
    }
    for (int i = 0; i < n; i++) {
        const leon_picture& p = pics[i].p;
        const int refs[2] = {p.type != LEON_PIC_I ? p.ref_fwd_slot : -1, p.type == LEON_PIC_B ? p.ref_bwd_slot : -1};
        for (int r : refs)
            if (r >= 0 && d->writer_epoch[r] == d->epoch)
                return fail(LEON_ERR_INVALID, "pictures %d and %d of one batch depend on each other (picture %d reads slot %d)",
                            d->writer_of[r], i, i, r);
    }
    return LEON_OK;
}

int submit_batch_any(leon_decoder* d, const AnyPic* pics, int n, int mem)
{
    HIP_TRY(hipSetDevice(d->dev));
    int rc = check_batch(d, pics, n);
    if (rc != LEON_OK) return rc;
    if (mem == LEON_MEM_HOST) {
        for (int i = 0; i < n; i++) {
            rc = submit_picture_any(d, pics[i]);
            if (rc != LEON_OK) return rc;
        }
        return LEON_OK;
    }
    int at = 0;
    rc = reserve_descs(d, n, at);
    if (rc != LEON_OK) return rc;
    {
        std::vector<int32_t> outs(n);
        for (int i = 0; i < n; i++) outs[i] = pics[i].p.out_slot;
        rc = guard_pending_conversions(d, outs.data(), n);
        if (rc != LEON_OK) return rc;
    }
    int count[6];
    uint64_t entries[6];
    sorted_descs(d, pics, n, d->h_desc_pinned + at, count, entries);
    double bytes[6] = {0, 0, 0, 0, 0, 0};
    if (d->timing)   // a measurement mode: the maps are read back to price the launches
        for (int i = 0; i < n; i++) {
            double b = 0;
            rc = algo_bytes_of(d, pics[i].p, true, b);
            if (rc != LEON_OK) return rc;
            bytes[class_of(pics[i].p)] += b;
        }
    HIP_TRY(hipMemcpyAsync(d->d_desc_ring + at, d->h_desc_pinned + at, sizeof(PicDesc) * n, hipMemcpyHostToDevice, d->stream));
    rc = launch_recon(d, d->d_desc_ring + at, count, bytes, pics[0].sparse, entries);
    if (rc != LEON_OK) return rc;
    return commit_descs(d, at, n);
}

int batch_create_any(leon_decoder* d, const AnyPic* pics, int n, leon_batch** out)
{
    HIP_TRY(hipSetDevice(d->dev));
    *out = nullptr;
    std::vector<PicDesc> h(n);
    int rc = check_batch(d, pics, n);
    if (rc != LEON_OK) return rc;
    leon_batch* b = new (std::nothrow) leon_batch();
    if (!b) return fail(LEON_ERR_NOMEM, "out of host memory");
    b->n = n;
    b->sparse = pics[0].sparse;
    sorted_descs(d, pics, n, h.data(), b->count, b->entries_of_type);
    for (int i = 0; i < n; i++) {
        double pb = 0;
        rc = algo_bytes_of(d, pics[i].p, true, pb);
        if (rc != LEON_OK) {
            delete b;
            return rc;
        }
        b->bytes_of_type[class_of(pics[i].p)] += pb;
    }
    b->out_slots.resize(n);
    for (int i = 0; i < n; i++) b->out_slots[i] = pics[i].p.out_slot;
    if (hipMalloc(&b->d_descs, sizeof(PicDesc) * n) != hipSuccess) {
        delete b;
        return fail(LEON_ERR_NOMEM, "descriptor allocation failed");
    }
    hipError_t e = hipMemcpy(b->d_descs, h.data(), sizeof(PicDesc) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(b->d_descs);
        delete b;
        return fail(LEON_ERR_HIP, "descriptor upload: %s", hipGetErrorString(e));
    }
    *out = b;
    return LEON_OK;
}

template <typename P>
std::vector<AnyPic> wrap(const P* pics, int n)
{
    std::vector<AnyPic> v((size_t)n);
    for (int i = 0; i < n; i++) v[(size_t)i] = any_of(pics[i]);
    return v;
}

}  // namespace

extern "C" {

int leon_submit_batch(leon_decoder* d, const leon_picture* pics, int32_t n, int32_t mem)
{
    if (!d || !pics || n <= 0) return fail(LEON_ERR_INVALID, "bad batch");
    return submit_batch_any(d, wrap(pics, n).data(), n, mem);
}

int leon_submit_sparse(leon_decoder* d, const leon_sparse_picture* pics, int32_t n, int32_t mem)
{
    if (!d || !pics || n <= 0) return fail(LEON_ERR_INVALID, "bad batch");
    return submit_batch_any(d, wrap(pics, n).data(), n, mem);
}

int leon_submit_picture(leon_decoder* d, const leon_picture* pic)
{
    if (!d || !pic) return fail(LEON_ERR_INVALID, "null argument");
    return submit_picture_any(d, any_of(*pic));
}

int leon_batch_create(leon_decoder* d, const leon_picture* pics, int32_t n, leon_batch** out)
{
    if (!d || !pics || n <= 0 || !out) return fail(LEON_ERR_INVALID, "bad batch");
    return batch_create_any(d, wrap(pics, n).data(), n, out);
}

int leon_batch_create_sparse(leon_decoder* d, const leon_sparse_picture* pics, int32_t n, leon_batch** out)
{
    if (!d || !pics || n <= 0 || !out) return fail(LEON_ERR_INVALID, "bad batch");
    return batch_create_any(d, wrap(pics, n).data(), n, out);
}

int leon_batch_run(leon_decoder* d, const leon_batch* b)
{
    if (!d || !b) return fail(LEON_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(d->dev));
    int rc = guard_pending_conversions(d, b->out_slots.data(), b->n);
    if (rc != LEON_OK) return rc;
    return launch_recon(d, b->d_descs, b->count, b->bytes_of_type, b->sparse, b->entries_of_type);
}

void leon_batch_destroy(leon_decoder* d, leon_batch* b)
{
    if (!b) return;
    if (d) {
        hipSetDevice(d->dev);
        hipStreamSynchronize(d->stream);
    }
    if (b->d_descs) hipFree(b->d_descs);
    delete b;
}

int leon_convert_rgba_batch(leon_decoder* d, const int32_t* slots, int32_t n, void* rgba_device, int32_t flavour)
{
    if (!d || !slots || n <= 0 || !rgba_device) return fail(LEON_ERR_INVALID, "bad argument");
    if (flavour != LEON_RGB_CPU_TWIN && flavour != LEON_RGB_GL) return fail(LEON_ERR_INVALID, "flavour %d", flavour);
    if (n > leon_decoder::kSlotIdRing) return fail(LEON_ERR_INVALID, "more than %d frames in one conversion", leon_decoder::kSlotIdRing);
    HIP_TRY(hipSetDevice(d->dev));
    for (int i = 0; i < n; i++)
        if (slots[i] < 0 || slots[i] >= d->cfg.n_slots) return fail(LEON_ERR_INVALID, "slot %d", slots[i]);
    if (d->slot_id_head + n > leon_decoder::kSlotIdRing) {
        HIP_TRY(hipStreamSynchronize(d->stream));
        if (d->conv_stream) HIP_TRY(hipStreamSynchronize(d->conv_stream));
        d->slot_id_head = 0;
    }
    int at = d->slot_id_head;
    d->slot_id_head += n;
    memcpy(d->h_slot_ids + at, slots, sizeof(int32_t) * n);
    hipStream_t cs = d->stream;
    if (d->overlap_convert) {      // second stream, ordered behind the reconstruction submitted so far
        cs = d->conv_stream;
        HIP_TRY(hipEventRecord(d->ev_recon_done, d->stream));
        HIP_TRY(hipStreamWaitEvent(cs, d->ev_recon_done, 0));
    }
    HIP_TRY(hipMemcpyAsync(d->d_slot_ids + at, d->h_slot_ids + at, sizeof(int32_t) * n, hipMemcpyHostToDevice, cs));
    RgbaGeom G{};
    G.cw = d->cfg.coded_width;
    G.ch = d->cfg.coded_height;
    G.fw = d->cfg.frame_width;
    G.fh = d->cfg.frame_height;
    G.cols = G.fw >> 1;
    G.rows = G.fh >> 1;
    G.n = n;
    G.flavour = flavour;
    G.slot_stride_lo = (uint32_t)(d->slot_stride & 0xffffffffu);
    G.slot_stride_hi = (uint32_t)(d->slot_stride >> 32);
    {
        const uint32_t c4 = (uint32_t)G.fw >> 2;
        G.inv_cols4 = c4 <= 1 ? 0u : (uint32_t)(((1ull << 32) + c4 - 1) / c4);
    }
    TimedLaunch tl{};
    if (d->timing) {
        if (get_event_pair(d, tl.a, tl.b) != LEON_OK) return LEON_ERR_HIP;
        tl.kind = 1;
        tl.mbs = (uint64_t)d->geom.mbw * d->geom.mbh * n;
        tl.bytes = kRgbaBytesPerMb * (double)tl.mbs;
        HIP_TRY(hipEventRecord(tl.a, cs));
    }
    if (flavour == LEON_RGB_CPU_TWIN) {
        if ((G.fw & 1) || (G.fh & 1)) {   // bytes the quad loop never reaches stay 255 (fillArray)
            size_t nd = (size_t)G.fw * G.fh * n;
            hipLaunchKernelGGL(k_fill255, dim3((unsigned)((nd + kRgbaBlock - 1) / kRgbaBlock)), dim3(kRgbaBlock), 0, cs, (uint32_t*)rgba_device, nd);
        }
        // k_rgba_twin4 divides by multiply-high, exact while idx * cols4 < 2^32 (idx < cols4 * rows);
        // larger frames take the generic kernel (leon_create caps the coded size at 4096, which stays inside)
        const uint64_t c4 = (uint64_t)(G.fw >> 2);
        const bool twin4_exact = c4 * c4 * (uint64_t)G.rows < (1ull << 32);
        if (G.cols > 0 && G.rows > 0 && (G.fw & 3) == 0 && ((size_t)rgba_device & 15) == 0 && twin4_exact)
            hipLaunchKernelGGL(k_rgba_twin4, dim3(((unsigned)(G.fw / 4) * (unsigned)G.rows + kRgbaBlock - 1) / kRgbaBlock, 1, n), dim3(kRgbaBlock), 0, cs,
                               d->d_slots, d->d_slot_ids + at, (uint8_t*)rgba_device, G);
        else if (G.cols > 0 && G.rows > 0)
            hipLaunchKernelGGL(k_rgba_twin, dim3((G.cols + kRgbaBlock - 1) / kRgbaBlock, G.rows, n), dim3(kRgbaBlock), 0, cs,
                               d->d_slots, d->d_slot_ids + at, (uint8_t*)rgba_device, G);
    } else {
        hipLaunchKernelGGL(k_rgba_gl, dim3((G.fw + kRgbaBlock - 1) / kRgbaBlock, G.fh, n), dim3(kRgbaBlock), 0, cs,
                           d->d_slots, d->d_slot_ids + at, (uint8_t*)rgba_device, G);
    }
    if (d->geom.alpha) {
        // yuva: A bytes from the fourth plane, over what the conversion wrote (the CPU twin's quad loop covers
        // whole quads only; the GL flavour every pixel)
        const int cover_w = flavour == LEON_RGB_CPU_TWIN ? (G.fw >> 1) * 2 : G.fw, cover_h = flavour == LEON_RGB_CPU_TWIN ? (G.fh >> 1) * 2 : G.fh;
        if (cover_w > 0 && cover_h > 0)
            hipLaunchKernelGGL(k_rgba_alpha, dim3((cover_w / 2 + kRgbaBlock - 1) / kRgbaBlock, cover_h, n), dim3(kRgbaBlock), 0, cs,
                               d->d_slots, d->d_slot_ids + at, (uint8_t*)rgba_device, G, cover_w, cover_h);
    }
    HIP_TRY(hipGetLastError());
    if (d->timing) {
        HIP_TRY(hipEventRecord(tl.b, cs));
        d->timed.push_back(tl);
    }
    if (d->overlap_convert) {
        HIP_TRY(hipEventRecord(d->ev_conv_done, cs));
        for (int i = 0; i < n; i++) d->conv_pending[slots[i]] = 1;
    }
    return LEON_OK;
}

int leon_convert_rgba(leon_decoder* d, int32_t slot, void* rgba, int32_t dst_mem, int32_t flavour)
{
    if (!d || !rgba) return fail(LEON_ERR_INVALID, "null argument");
    if (dst_mem == LEON_MEM_DEVICE) return leon_convert_rgba_batch(d, &slot, 1, rgba, flavour);
    HIP_TRY(hipSetDevice(d->dev));
    size_t bytes = (size_t)d->cfg.frame_width * d->cfg.frame_height * 4;
    if (!d->d_rgba_tmp) HIP_TRY(hipMalloc(&d->d_rgba_tmp, bytes));
    const bool ov = d->overlap_convert;
    d->overlap_convert = false;                 // host destination: plain in-order path
    int rc = leon_convert_rgba_batch(d, &slot, 1, d->d_rgba_tmp, flavour);
    d->overlap_convert = ov;
    if (rc != LEON_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgba, d->d_rgba_tmp, bytes, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LEON_OK;
}

int leon_read_planes(leon_decoder* d, int32_t slot, uint8_t* y, uint8_t* cb, uint8_t* cr)
{
    if (!d || slot < 0 || slot >= d->cfg.n_slots) return fail(LEON_ERR_INVALID, "slot %d", slot);
    HIP_TRY(hipSetDevice(d->dev));
    size_t ny = (size_t)d->geom.cw * d->geom.ch, nc = ny >> 2;
    const uint8_t* base = d->d_slots + (size_t)slot * d->slot_stride;
    if (y) HIP_TRY(hipMemcpyAsync(y, base, ny, hipMemcpyDeviceToHost, d->stream));
    if (cb) HIP_TRY(hipMemcpyAsync(cb, base + ny, nc, hipMemcpyDeviceToHost, d->stream));
    if (cr) HIP_TRY(hipMemcpyAsync(cr, base + ny + nc, nc, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LEON_OK;
}

int leon_write_planes(leon_decoder* d, int32_t slot, const uint8_t* y, const uint8_t* cb, const uint8_t* cr)
{
    if (!d || slot < 0 || slot >= d->cfg.n_slots) return fail(LEON_ERR_INVALID, "slot %d", slot);
    HIP_TRY(hipSetDevice(d->dev));
    size_t ny = (size_t)d->geom.cw * d->geom.ch, nc = ny >> 2;
    uint8_t* base = d->d_slots + (size_t)slot * d->slot_stride;
    if (y) HIP_TRY(hipMemcpyAsync(base, y, ny, hipMemcpyHostToDevice, d->stream));
    if (cb) HIP_TRY(hipMemcpyAsync(base + ny, cb, nc, hipMemcpyHostToDevice, d->stream));
    if (cr) HIP_TRY(hipMemcpyAsync(base + ny + nc, cr, nc, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LEON_OK;
}

int leon_read_alpha_plane(leon_decoder* d, int32_t slot, uint8_t* a)
{
    if (!d || slot < 0 || slot >= d->cfg.n_slots || !a) return fail(LEON_ERR_INVALID, "slot %d", slot);
    if (!d->geom.alpha) return fail(LEON_ERR_INVALID, "not a yuva decoder");
    HIP_TRY(hipSetDevice(d->dev));
    const size_t ny = (size_t)d->geom.cw * d->geom.ch;
    HIP_TRY(hipMemcpyAsync(a, d->d_slots + (size_t)slot * d->slot_stride + ny + ny / 2, ny, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LEON_OK;
}

int leon_write_alpha_plane(leon_decoder* d, int32_t slot, const uint8_t* a)
{
    if (!d || slot < 0 || slot >= d->cfg.n_slots || !a) return fail(LEON_ERR_INVALID, "slot %d", slot);
    if (!d->geom.alpha) return fail(LEON_ERR_INVALID, "not a yuva decoder");
    HIP_TRY(hipSetDevice(d->dev));
    const size_t ny = (size_t)d->geom.cw * d->geom.ch;
    HIP_TRY(hipMemcpyAsync(d->d_slots + (size_t)slot * d->slot_stride + ny + ny / 2, a, ny, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LEON_OK;
}

int leon_slot_device_ptr(leon_decoder* d, int32_t slot, void** ptr, size_t* bytes)
{
    if (!d || slot < 0 || slot >= d->cfg.n_slots || !ptr) return fail(LEON_ERR_INVALID, "slot %d", slot);
    *ptr = d->d_slots + (size_t)slot * d->slot_stride;
    if (bytes) *bytes = d->plane_bytes;
    return LEON_OK;
}

int leon_sync(leon_decoder* d)
{
    if (!d) return fail(LEON_ERR_INVALID, "null decoder");
    HIP_TRY(hipSetDevice(d->dev));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (d->conv_stream) {
        HIP_TRY(hipStreamSynchronize(d->conv_stream));
        std::fill(d->conv_pending.begin(), d->conv_pending.end(), 0);
    }
    return LEON_OK;
}

int leon_set_overlap_convert(leon_decoder* d, int32_t on)
{
    if (!d) return fail(LEON_ERR_INVALID, "null decoder");
    HIP_TRY(hipSetDevice(d->dev));
    if (on && !d->conv_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&d->conv_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&d->ev_recon_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&d->ev_conv_done, hipEventDisableTiming));
        d->conv_pending.assign(d->cfg.n_slots, 0);
    }
    if (!on && d->conv_stream) HIP_TRY(hipStreamSynchronize(d->conv_stream));
    d->overlap_convert = on != 0;
    return LEON_OK;
}

int leon_timing_enable(leon_decoder* d, int32_t on)
{
    if (!d) return fail(LEON_ERR_INVALID, "null decoder");
    d->timing = on != 0;
    if (d->timing && d->ev_pool.size() < 4096) {
        // events for ~2000 launches up front: creating them one by one would sit inside the caller's
        // timed region
        HIP_TRY(hipSetDevice(d->dev));
        while (d->ev_pool.size() < 4096) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            d->ev_pool.push_back(e);
        }
    }
    return LEON_OK;
}

int leon_timing_reset(leon_decoder* d)
{
    if (!d) return fail(LEON_ERR_INVALID, "null decoder");
    HIP_TRY(hipSetDevice(d->dev));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (d->conv_stream) HIP_TRY(hipStreamSynchronize(d->conv_stream));
    for (auto& t : d->timed) {
        d->ev_pool.push_back(t.a);
        d->ev_pool.push_back(t.b);
    }
    d->timed.clear();
    return LEON_OK;
}

int leon_timing_get(leon_decoder* d, int32_t kind, leon_kernel_stats* out)
{
    if (!d || !out) return fail(LEON_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(d->dev));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (d->conv_stream) HIP_TRY(hipStreamSynchronize(d->conv_stream));
    leon_kernel_stats s{};
    for (auto& t : d->timed) {
        // kinds 2..4: the reconstruction launches of one picture type (I, P, B)
        if (kind >= 2 ? (t.kind != 0 || t.pic_type != kind - 1) : t.kind != kind) continue;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        s.launches++;
        s.total_ms += ms;
        s.algorithmic_bytes += t.bytes;
        s.macroblocks += t.mbs;
    }
    *out = s;
    return LEON_OK;
}

int leon_timing_get_launches(leon_decoder* d, leon_launch_time* out, int32_t cap, int32_t* n)
{
    if (!d || !n || (cap > 0 && !out)) return fail(LEON_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(d->dev));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (d->conv_stream) HIP_TRY(hipStreamSynchronize(d->conv_stream));
    *n = (int32_t)d->timed.size();
    for (size_t i = 0; i < d->timed.size() && (int32_t)i < cap; i++) {
        const TimedLaunch& t = d->timed[i];
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        out[i] = leon_launch_time{t.kind, t.pic_type, (double)ms, t.bytes, t.mbs};
    }
    return LEON_OK;
}

int leon_device_malloc(int32_t device_id, size_t bytes, void** ptr, int32_t* contiguous)
{
    if (!ptr || bytes == 0) return fail(LEON_ERR_INVALID, "null pointer or zero bytes");
    *ptr = nullptr;
    HIP_TRY(hipSetDevice(device_id));
    bool c = false;
    if (big_alloc(ptr, bytes, kBigCaller, &c) != hipSuccess) {
        (void)hipGetLastError();
        return fail(LEON_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
    }
    if (contiguous) *contiguous = c ? 1 : 0;
    return LEON_OK;
}

int leon_device_free(void* ptr)
{
    if (!ptr) return LEON_OK;
    // like hipFree, this waits for the device: a caller may free a buffer its last launch is still writing.  A buffer
    // of the contiguous pool is then handed to the next request instead of to the driver (big_alloc says why).
    const int pooled_on = contig_pool().device_of(ptr);
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (pooled_on >= 0 && pooled_on != cur) HIP_TRY(hipSetDevice(pooled_on));
    const hipError_t e = pooled_on >= 0 ? hipDeviceSynchronize() : hipSuccess;       // (hipFree below waits by itself)
    if (pooled_on >= 0 && pooled_on != cur) (void)hipSetDevice(cur);
    HIP_TRY(e);
    HIP_TRY(big_free(ptr));
    return LEON_OK;
}

int leon_device_pool_stats(uint64_t* held_bytes, uint64_t* in_use_bytes, int32_t* segments)
{
    ContigPool& P = contig_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (held_bytes) *held_bytes = P.held;
    if (in_use_bytes) *in_use_bytes = P.in_use;
    if (segments) *segments = P.n_segments;
    return LEON_OK;
}

int leon_measure_copy_bandwidth(leon_decoder* d, size_t bytes, int32_t iters, double* gbps)
{
    if (!d || !gbps || bytes < 4096 || iters < 1) return fail(LEON_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(d->dev));
    bytes &= ~(size_t)4095;
    uint4 *src = nullptr, *dst = nullptr;
    if (big_alloc((void**)&src, bytes, kBigCaller) != hipSuccess) return fail(LEON_ERR_NOMEM, "copy source");      // allocated like the buffers it is the yardstick for
    if (big_alloc((void**)&dst, bytes, kBigCaller) != hipSuccess) {
        big_free(src);
        return fail(LEON_ERR_NOMEM, "copy destination");
    }
    hipMemsetAsync(src, 0x5a, bytes, d->stream);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    size_t n = bytes / 16;
    unsigned grid = (unsigned)((n + kRgbaBlock - 1) / kRgbaBlock);
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(kRgbaBlock), 0, d->stream, src, dst, n);   // warm-up
    hipEventRecord(a, d->stream);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(kRgbaBlock), 0, d->stream, src, dst, n);
    hipEventRecord(b, d->stream);
    hipError_t e = hipStreamSynchronize(d->stream);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    // the stream is idle; contiguous buffers return to the pool (the next call, or a decoder's ring, takes them) -- the
    // yardstick frees nothing the copy kernel wrote in front of the pipelines bench.py runs afterwards
    big_free(src);
    big_free(dst);
    if (e != hipSuccess) return fail(LEON_ERR_HIP, "copy kernel: %s", hipGetErrorString(e));
    *gbps = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
    return LEON_OK;
}

int leon_measure_stream_bandwidth(leon_decoder* d, size_t bytes, int32_t iters, int32_t reads, int32_t writes, double* gbps)
{
    if (!d || !gbps || bytes < 4096 || iters < 1 || reads < 0 || reads > 2 || writes < 0 || writes > 2 || reads + writes == 0)
        return fail(LEON_ERR_INVALID, "bad argument (0..2 read streams, 0..2 write streams, not both 0)");
    HIP_TRY(hipSetDevice(d->dev));
    bytes &= ~(size_t)4095;
    uint4* buf[4] = {nullptr, nullptr, nullptr, nullptr};       // s0 s1 d0 d1
    const bool want[4] = {reads >= 1, reads >= 2, true, writes >= 2};      // d0 always: the read-only kernel names it
    auto release = [&] { for (uint4*& b : buf) { if (b) big_free(b); b = nullptr; } };
    for (int k = 0; k < 4; k++)
        if (want[k] && big_alloc((void**)&buf[k], bytes, kBigCaller) != hipSuccess) { release(); return fail(LEON_ERR_NOMEM, "stream buffer %d", k); }
    for (int k = 0; k < 2; k++) if (buf[k]) hipMemsetAsync(buf[k], 0x5a, bytes, d->stream);
    const size_t n = bytes / 16;
    const dim3 grid((unsigned)((n + kRgbaBlock - 1) / kRgbaBlock)), block(kRgbaBlock);
    auto launch = [&] {
        switch (reads * 3 + writes) {
        case 1: hipLaunchKernelGGL((k_stream16<0, 1>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        case 2: hipLaunchKernelGGL((k_stream16<0, 2>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        case 3: hipLaunchKernelGGL((k_stream16<1, 0>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        case 4: hipLaunchKernelGGL((k_stream16<1, 1>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        case 5: hipLaunchKernelGGL((k_stream16<1, 2>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        case 6: hipLaunchKernelGGL((k_stream16<2, 0>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        case 7: hipLaunchKernelGGL((k_stream16<2, 1>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        default: hipLaunchKernelGGL((k_stream16<2, 2>), grid, block, 0, d->stream, buf[0], buf[1], buf[2], buf[3], n); break;
        }
    };
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();                                   // warm-up
    hipEventRecord(a, d->stream);
    for (int i = 0; i < iters; i++) launch();
    hipEventRecord(b, d->stream);
    const hipError_t e = hipStreamSynchronize(d->stream);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    release();                                  // the stream is idle; pooled buffers stay with the process (big_alloc)
    if (e != hipSuccess) return fail(LEON_ERR_HIP, "stream kernel: %s", hipGetErrorString(e));
    *gbps = (double)(reads + writes) * (double)bytes * iters / (ms * 1e-3) / 1e9;
    return LEON_OK;
}

}  // extern "C"

// the native decode pipeline (include/leon_pipeline.h) lives in the same translation unit: it drives
// the decoder through the internals above
#include "leon_pipeline_impl.h"
