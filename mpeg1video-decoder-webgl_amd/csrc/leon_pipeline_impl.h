// leon_pipeline_impl.h -- implementation of include/leon_pipeline.h; included at the end of leon_hip.cpp
// (it uses the decoder's internals: submit_batch_any, the slot ring, the decoder's stream).
#include "../../include/leon_pipeline.h"
#include "../../include/leon_vlc.h"
#include "leon_vlc_gpu.h"

#include <atomic>
#include <sys/stat.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

namespace {

using Clock = std::chrono::steady_clock;

struct PipePic {                 // one parsed picture: offsets of its arrays inside the GOP's arena
    int32_t type, tref;
    uint32_t n_entries;
    double ts_ms;
    size_t grp_off, entries, qscale, intra, repadd, mb_dir, mv_fwd, mv_bwd;   // (size_t)-1: absent
    int32_t qm = -1;             // index into its GOP's `qms` (matrices of the sequence header in force), -1: the stream's first
};

struct Arena {                   // pinned host buffer + its device twin, one GOP at a time
    char* host = nullptr;
    char* dev = nullptr;
    size_t cap = 0, host_cap = 0, used = 0;      // cap: of the device twin
    bool host_owned = true, dev_owned = true;    // false: a piece of the pipeline's slabs (gpu_parser), not to be freed
};

struct GopJob {
    bool open_gop = false;       // its GOP header says closed_gop = 0
    uint64_t gop = 0;            // running index among the GOPs of this pipeline
    uint64_t key_gop = 0;        // GOP id in the stream (key-map index, counting on across loops)
    Arena* arena = nullptr;
    std::vector<PipePic> pics;
    double gop_ts_ms = 0;
    int status = LEON_OK;
    std::string err;
    // sequence headers of this shard whose quantiser matrices differ from the stream's first: the reference reloads them
    // at every header (decoders/jsv.js:540-558), a picture is dequantised with the ones in force when it was coded
    std::vector<std::array<uint8_t, 128>> qms;
    // gpu_parser: the slices of the GOP and its pictures as the device kernels want them (pointers into the arena's
    // device twin; VlcSlice::pic counts inside the GOP until submit_window rebases it), what of the arena is uploaded
    // (stream bytes) and what is cleared on the device (counters, error words, maps)
    std::vector<leon::VlcSlice> slices;
    std::vector<leon::VlcPic> vpics;
    size_t upload_bytes = 0, zero_begin = 0, zero_bytes = 0;
};

struct PipeWindow {
    int64_t id = 0;
    int ring = 0;
    std::vector<GopJob*> jobs;
    std::vector<leon_pipeline_frame> frames;
    hipEvent_t done = nullptr;
    int status = LEON_OK;
    uint32_t n_vpics = 0;        // gpu_parser: error words of the window's pictures are in the ring entry's h_err
};

struct VlcRing {                 // gpu_parser: per ring entry, the window's slice / picture descriptors
    char* h = nullptr;           // pinned: [VlcSlice x n][VlcPic x m]
    char* d = nullptr;           // device: the same + [slice_words x n][error x m]
    size_t cap = 0;
    uint32_t* h_err = nullptr;   // pinned copy of the error words
    size_t err_cap = 0;
};

constexpr size_t kNone = (size_t)-1;
inline size_t pad256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

struct leon_pipeline {
    leon_pipeline_config cfg{};
    leon_pipeline_info info{};
    const uint8_t* stream = nullptr;
    size_t bytes = 0;
    size_t valid = 0;            // how much of the stream has arrived (leon_pipeline_feed); under mu
    leon_pipeline_callback cb = nullptr;
    void* user = nullptr;
    leon_vlc_info vinfo{};
    std::vector<uint64_t> shard_begin, shard_end;     // byte ranges of the GOP shards
    std::vector<uint32_t> mine;                       // key-map ids this pipeline decodes (all, or g % shard_count == shard_index)
    uint64_t total_gops = 0;                          // gops * loop
    int W = 32, R = 2, K = 1, max_pics = 16;
    std::vector<double> st_window_done;                      // LEON_DEBUG_PIPE_TIMING: when each window completed (seconds from the start)
    uint64_t st_wait_ring_ns = 0, st_wait_scan_ns = 0;       // submit thread: waiting for a ring entry / for the window's GOPs to be parsed
    std::string capture_dir;     // LEON_DEBUG_CAPTURE=<dir> at create: everything a window's launches read and wrote goes to files (capture_*)
    bool unfused = false;        // frame_width % 8 != 0, or the GL flavour: planes for every picture, one display conversion launch per picture
    int flavour = LEON_RGB_CPU_TWIN;      // leon_pipeline_config.display_flavour
    size_t frame_bytes = 0;

    leon_decoder* dec = nullptr;
    hipStream_t copy_stream = nullptr;
    uint8_t* d_rgba = nullptr;                        // R ring entries of W * max_pics frames
    bool gpu_parser = false;
    // The parser kernels of window n + 1 run beside the reconstruction of window n -- and beside the parser kernels of
    // window n + 2, on a second stream: a parse launch lasts as long as its longest slice (one lane, symbol after symbol) and
    // leaves three quarters of the issue slots idle; two of them side by side take little longer than one.
    hipStream_t vlc_stream[2] = {nullptr, nullptr};
    leon::VlcTables* d_vlc_tables = nullptr;
    leon::VlcGeom vgeom{};
    char* slab_host = nullptr;        // gpu_parser: the arenas' memory, one allocation each (pinned host, device)
    char* slab_dev = nullptr;
    size_t vlc_index_lds = 0;         // dynamic LDS of k_vlc_index: the group counters of one picture + its scan
    std::vector<VlcRing> vlc_ring;

    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> next_gop{0};
    std::map<uint64_t, GopJob*> parsed;               // waiting for the submit thread
    std::deque<Arena*> free_arenas;
    std::deque<PipeWindow*> to_notify;
    std::vector<int64_t> ring_owner;                  // window id holding a ring entry, -1 free
    std::map<int64_t, PipeWindow*> delivered;         // waiting for release
    int64_t windows_submitted = 0, windows_done = 0, total_windows = 0;
    bool stop = false, finished = false, quiet = false, submit_exited = false;
    int status = LEON_OK;
    std::string err;

    std::vector<std::thread> parsers;
    std::thread submitter, notifier;
    std::vector<Arena*> all_arenas;

    // stats
    Clock::time_point t0;
    std::atomic<uint64_t> st_pictures{0}, st_gops{0}, st_entries{0}, st_parse_ns{0}, st_upload{0}, st_submit_ns{0}, st_submit_wait_ns{0};
    double st_seconds = 0;
};

namespace {

void pipe_fail(leon_pipeline* p, int code, const std::string& msg)
{
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->status == LEON_OK) {
        p->status = code;
        p->err = msg;
    }
    p->stop = true;
    p->cv.notify_all();
}

// host_need bytes of pinned memory and dev_need bytes of its device twin (equal for a GOP parsed on the host: the
// arena is copied as it is; the GPU parser uploads the stream bytes only and keeps everything else on the device)
bool arena_reserve(leon_pipeline* p, Arena* a, size_t host_need, size_t dev_need)
{
    (void)p;
    if (host_need > a->host_cap) {
        const size_t cap = std::max(host_need + host_need / 2, (size_t)4 << 20);
        char* h = nullptr;
        if (hipHostMalloc((void**)&h, cap, hipHostMallocDefault) != hipSuccess) return false;
        if (a->used) memcpy(h, a->host, std::min(a->used, a->host_cap));
        if (a->host && a->host_owned) hipHostFree(a->host);
        a->host = h;
        a->host_cap = cap;
        a->host_owned = true;
    }
    if (dev_need > a->cap) {
        const size_t cap = std::max(dev_need + dev_need / 2, (size_t)8 << 20);
        char* dv = nullptr;
        if (big_alloc((void**)&dv, cap, kBigArenas) != hipSuccess) return false;
        if (a->dev && a->dev_owned) big_free(a->dev);        // only ever grown while the arena is being filled: nothing in flight reads it
        a->dev = dv;
        a->cap = cap;
        a->dev_owned = true;
    }
    return true;
}

inline Arena* a_of(GopJob* job) { return job->arena; }

// The pipeline takes the picture size from the stream's FIRST sequence header (leon_create).  Every key-map entry starts
// with a sequence header of its own and the reference re-initialises at each (decoders/jsv.js:491-561): a shard whose
// header names another SIZE cannot be decoded into this pipeline's rings and is refused; other quantiser MATRICES
// (jsv.js:540-558) are carried with the shard's pictures and dequantised with (leon_add_quant_matrices, round 4 --
// round 3 refused such a stream).  Returns the index of the matrices in force in job->qms, -1 = the stream's first,
// -2 = refused (job->status / err set).
int sequence_in_force(leon_pipeline* p, GopJob* job, leon_vlc_stream* st, uint64_t g)
{
    leon_vlc_info v{};
    leon_vlc_get_info(st, &v);
    if (v.coded_width != p->vinfo.coded_width || v.coded_height != p->vinfo.coded_height ||
        v.frame_width != p->vinfo.frame_width || v.frame_height != p->vinfo.frame_height) {
        job->status = LEON_ERR_INVALID;
        job->err = std::string("GOP shard ") + std::to_string(g) + ": its sequence header changes the picture size (" +
                   std::to_string(v.frame_width) + "x" + std::to_string(v.frame_height) + " after " + std::to_string(p->vinfo.frame_width) + "x" +
                   std::to_string(p->vinfo.frame_height) + "): a pipeline decodes one size";
        return -2;
    }
    if (memcmp(v.intra_qm, p->vinfo.intra_qm, 64) == 0 && memcmp(v.non_intra_qm, p->vinfo.non_intra_qm, 64) == 0) return -1;
    std::array<uint8_t, 128> m;
    memcpy(m.data(), v.intra_qm, 64);
    memcpy(m.data() + 64, v.non_intra_qm, 64);
    for (size_t i = 0; i < job->qms.size(); i++)
        if (job->qms[i] == m) return (int)i;
    job->qms.push_back(m);
    return (int)job->qms.size() - 1;
}

// gpu_parser: the host reads the picture layer only (leon_vlc_scan_picture) and lays the GOP's arena out for the
// device kernels of leon_vlc_gpu.h:
//   [stream bytes, zero padded]                                            uploaded
//   per picture: [macroblock records]                                      cleared on the device
//   per picture: [maps | grp_off | entries (capacity)], per slice: [block records]      written by the kernels
void scan_gop_for_gpu(leon_pipeline* p, GopJob* job, leon_vlc_stream* st, const uint8_t* bytes, size_t n, uint64_t g, int qm_now)
{
    struct Scan { leon_vlc_picture_scan s; std::vector<int32_t> code; std::vector<uint64_t> pos; int qm; };
    std::vector<Scan> scans;
    for (;;) {
        leon_vlc_picture_scan sc;
        const int rc = leon_vlc_scan_picture(st, &sc);
        if (rc == LEON_VLC_END) break;
        if (rc != LEON_VLC_PICTURE) {
            job->status = LEON_ERR_INVALID;
            job->err = std::string("GOP shard ") + std::to_string(g) + ": " + leon_vlc_last_error();
            return;
        }
        if (sc.new_sequence && (qm_now = sequence_in_force(p, job, st, g)) == -2) return;
        if (sc.open_gop) job->open_gop = true;
        Scan x;
        x.s = sc;
        x.qm = qm_now;
        x.code.assign(sc.slice_code, sc.slice_code + sc.n_slices);
        x.pos.assign(sc.slice_bit_pos, sc.slice_bit_pos + sc.n_slices);
        scans.push_back(std::move(x));
    }
    if ((int)scans.size() > p->max_pics) {
        job->status = LEON_ERR_INVALID;
        job->err = "a GOP has " + std::to_string(scans.size()) + " pictures; raise max_gop_pictures (" + std::to_string(p->max_pics) + ")";
        return;
    }
    if (n >= ((size_t)1 << 28)) { job->status = LEON_ERR_INVALID; job->err = "GOP shard too large for the GPU parser"; return; }
    const size_t mbs = (size_t)p->vinfo.mb_width * p->vinfo.mb_height;
    const size_t mpad = pad256(mbs), vpad = pad256(mbs * 4), gpad = pad256(((size_t)p->vinfo.n_groups + 1) * 4);
    const size_t rpad = pad256(mbs * leon::kVlcMbRecBytes);
    // the block records of a picture: one array for all its slices (leon_vlc_gpu.h VlcSliceOut), a record per block at most
    const size_t recpad = pad256(mbs * (size_t)leon::vlc_blocks_per_mb(p->vinfo.has_alpha == 1) * leon::kVlcRecWords * 4);
    const size_t max_entries = (size_t)p->vinfo.coded_width * p->vinfo.coded_height * (p->vinfo.has_alpha == 1 ? 5 : 3) / 2;
    const size_t stream_pad = pad256(n + 16);
    // sizes first: the arena may move when it grows
    const size_t maps_per_pic = 4 * mpad + 2 * vpad;
    size_t need = stream_pad + scans.size() * (rpad + recpad + maps_per_pic);
    std::vector<size_t> ecap(scans.size());
    for (size_t k = 0; k < scans.size(); k++) {
        const Scan& x = scans[k];
        size_t pic_words = 0;
        for (size_t j = 0; j < x.code.size(); j++) {
            const uint64_t begin = x.pos[j] >> 3, end = j + 1 < x.code.size() ? (x.pos[j + 1] >> 3) - 4 : x.s.end_byte;
            const size_t nb = end > begin ? (size_t)(end - begin) : 0;
            // An entry is a coefficient symbol, and n of them in a block take 3 n + 1 bits at least: the shortest code is
            // '11s' (three bits; '1s', two, in first position only), and the end-of-block code (or an intra block's DC size
            // code) another two -- entries <= bits / 3.  (Round 3 reserved a dword per two bits.)  The kernels keep inside
            // the bound whatever the stream says (k_vlc_blocks, and the reconstruction reads the list through a buffer
            // resource of exactly this length).
            pic_words += (8 * nb) / 3 + 2;
        }
        ecap[k] = std::min(std::max(pic_words, (size_t)64), max_entries);
        need += gpad + pad256(ecap[k] * 4 + 4);
    }
    if (!arena_reserve(p, a_of(job), stream_pad, need)) { job->status = LEON_ERR_NOMEM; job->err = "staging allocation failed"; return; }
    Arena* a = job->arena;
    memcpy(a->host, bytes, n);
    memset(a->host + n, 0, stream_pad - n);
    job->upload_bytes = stream_pad;
    size_t at = stream_pad;
    job->zero_begin = at;                                // the macroblock records of all its pictures, side by side
    job->zero_bytes = scans.size() * rpad;
    char* dev = a->dev;
    auto take = [&](size_t bytes_) { const size_t o = at; at += bytes_; return o; };
    for (size_t k = 0; k < scans.size(); k++) {
        const Scan& x = scans[k];
        PipePic m{};
        m.type = x.s.type;
        m.tref = x.s.temporal_reference;
        m.ts_ms = x.s.ts_ms;
        m.qm = x.qm;
        leon::VlcPic v{};
        v.type = x.s.type;
        v.full_pel_fwd = x.s.full_pel_fwd; v.fwd_rsize = x.s.fwd_rsize;
        v.full_pel_bwd = x.s.full_pel_bwd; v.bwd_rsize = x.s.bwd_rsize;
        v.zbase = dev + take(rpad);
        job->pics.push_back(m);
        job->vpics.push_back(v);
    }
    for (size_t k = 0; k < scans.size(); k++) {
        const Scan& x = scans[k];
        PipePic& m = job->pics[k];
        leon::VlcPic& v = job->vpics[k];
        v.maps = dev + at;                               // [qscale | intra | repadd | mb_dir | mv_fwd | mv_bwd]: VlcGeom's offsets
        m.qscale = take(mpad); m.intra = take(mpad);
        const size_t ra = take(mpad), md = take(mpad), mf = take(vpad), mk = take(vpad);
        m.repadd = x.s.type != LEON_PIC_I ? ra : kNone;
        m.mb_dir = x.s.type == LEON_PIC_B ? md : kNone;
        m.mv_fwd = x.s.type != LEON_PIC_I ? mf : kNone;
        m.mv_bwd = x.s.type == LEON_PIC_B ? mk : kNone;
        m.grp_off = take(gpad);
        m.entries = take(pad256(ecap[k] * 4 + 4));
        m.n_entries = (uint32_t)ecap[k];                 // the bound of the device lists (the kernels keep inside it)
        v.grp_off = (uint32_t*)(dev + m.grp_off);
        v.entries = (uint32_t*)(dev + m.entries);
        v.entries_cap = (uint32_t)ecap[k];
        v.recs = (uint32_t*)(dev + take(recpad));
        for (size_t j = 0; j < x.code.size(); j++) {
            leon::VlcSlice sl{};
            sl.bytes = (const uint32_t*)dev;
            sl.n_dwords = (uint32_t)(stream_pad / 4);
            sl.n_bytes = (uint32_t)n;
            sl.bit_pos = (uint32_t)x.pos[j];
            sl.end_byte = (uint32_t)(j + 1 < x.code.size() ? (x.pos[j + 1] >> 3) - 4 : x.s.end_byte);
            sl.code = x.code[j];
            sl.pic = (uint32_t)k;
            job->slices.push_back(sl);
        }
    }
    a->used = at;
    if (!job->pics.empty()) job->gop_ts_ms = job->pics[0].ts_ms;
}

// one GOP shard: parse every picture of it straight into a pinned arena
void parse_gop(leon_pipeline* p, GopJob* job)
{
    // job->gop counts the GOPs this pipeline decodes; which key-map entry that is:
    const uint64_t g = p->mine[job->gop % p->mine.size()];
    job->key_gop = (job->gop / p->mine.size()) * p->shard_begin.size() + g;      // id in the whole (looped) stream
    const uint8_t* b = p->stream + p->shard_begin[g];
    const size_t n = (size_t)(p->shard_end[g] - p->shard_begin[g]);
    leon_vlc_stream* st = nullptr;
    // a stream without key map is one shard that still carries its container header
    // gpu_parser: only the picture layer is read here -- in place, when the stream goes on behind the shard
    const int rc_open = p->gpu_parser ? leon_vlc_open_scan(b, n, p->bytes - (size_t)p->shard_begin[g], p->vinfo.has_alpha, &st)
                                      : leon_vlc_open_shard(b, n, 1, p->vinfo.has_alpha, &st);
    if (rc_open != LEON_VLC_OK) {
        job->status = LEON_ERR_INVALID;
        job->err = std::string("GOP shard ") + std::to_string(g) + ": " + leon_vlc_last_error();
        return;
    }
    int qm_now = sequence_in_force(p, job, st, g);        // the shard's own sequence header (read by leon_vlc_open*)
    if (qm_now == -2) {
        leon_vlc_close(st);
        return;
    }
    const size_t mbs = (size_t)p->vinfo.mb_width * p->vinfo.mb_height;
    const size_t mpad = pad256(mbs), vpad = pad256(mbs * 4), gpad = pad256(((size_t)p->vinfo.n_groups + 1) * 4);
    Arena* a = job->arena;
    a->used = 0;
    if (p->gpu_parser) {
        scan_gop_for_gpu(p, job, st, b, n, g, qm_now);
        leon_vlc_close(st);
        return;
    }
    leon_vlc_picture pic;
    for (;;) {
        const int rc = leon_vlc_next_picture_sync(st, &pic);      // parsed here, on this thread: no second thread per shard
        if (rc == LEON_VLC_END) break;
        if (rc != LEON_VLC_PICTURE) {
            job->status = LEON_ERR_INVALID;
            job->err = std::string("GOP shard ") + std::to_string(g) + ": " + leon_vlc_last_error();
            break;
        }
        if (pic.new_sequence && (qm_now = sequence_in_force(p, job, st, g)) == -2) break;
        if (pic.open_gop) job->open_gop = true;
        const size_t epad = pad256((size_t)pic.n_entries * 4 + 4);
        const size_t need = a->used + gpad + epad + 4 * mpad + 2 * vpad;
        if (!arena_reserve(p, a, need, need)) {
            job->status = LEON_ERR_NOMEM;
            job->err = "pinned staging allocation failed";
            break;
        }
        PipePic m{};
        m.type = pic.type;
        m.tref = pic.temporal_reference;
        m.n_entries = pic.n_entries;
        m.ts_ms = pic.ts_ms;
        m.qm = qm_now;
        auto put = [&](const void* src, size_t nbytes, size_t reserve) -> size_t {
            if (!src) return kNone;
            const size_t at = a->used;
            memcpy(a->host + at, src, nbytes);
            a->used += reserve;
            return at;
        };
        m.grp_off = put(pic.grp_off, ((size_t)pic.n_groups + 1) * 4, gpad);
        m.entries = put(pic.entries ? (const void*)pic.entries : (const void*)pic.grp_off, (size_t)pic.n_entries * 4, epad);
        m.qscale = put(pic.qscale, mbs, mpad);
        m.intra = put(pic.intra, mbs, mpad);
        m.repadd = put(pic.type != LEON_PIC_I ? pic.repadd : nullptr, mbs, mpad);
        m.mb_dir = put(pic.type == LEON_PIC_B ? pic.mb_dir : nullptr, mbs, mpad);
        m.mv_fwd = put(pic.type != LEON_PIC_I ? pic.mv_fwd : nullptr, mbs * 4, vpad);
        m.mv_bwd = put(pic.type == LEON_PIC_B ? pic.mv_bwd : nullptr, mbs * 4, vpad);
        if (job->pics.empty()) job->gop_ts_ms = pic.ts_ms;
        job->pics.push_back(m);
        p->st_entries += pic.n_entries;
    }
    leon_vlc_close(st);
    if (job->status == LEON_OK && (int)job->pics.size() > p->max_pics) {
        job->status = LEON_ERR_INVALID;
        job->err = "a GOP has " + std::to_string(job->pics.size()) + " pictures; raise max_gop_pictures (" + std::to_string(p->max_pics) + ")";
    }
}

void parser_main(leon_pipeline* p)
{
    hipSetDevice(p->cfg.device_id);
    for (;;) {
        Arena* a = nullptr;
        uint64_t g;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            // an arena is the ticket to parse ahead: there are W * (R + 1) of them
            p->cv.wait(lk, [&] { return p->stop || !p->free_arenas.empty(); });
            if (p->stop) return;
            g = p->next_gop.load();
            if (g >= p->total_gops) return;
            p->next_gop = g + 1;
            a = p->free_arenas.front();
            p->free_arenas.pop_front();
        }
        GopJob* job = new GopJob();
        job->gop = g;
        job->arena = a;
        {   // a stream that is still arriving (leon_pipeline_create_partial): the GOP's bytes must be there -- the
            // decoder of the reference stalls the same way when its buffer runs dry (features/bitreader.js:135-189)
            const uint64_t need = p->shard_end[p->mine[g % p->mine.size()]];
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || p->valid >= need; });
            if (p->stop) { p->free_arenas.push_back(a); delete job; return; }
        }
        {
            const uint8_t* b = p->stream + p->shard_begin[p->mine[g % p->mine.size()]];
            if (b[0] != 0 || b[1] != 0 || b[2] != 1) {
                job->status = LEON_ERR_INVALID;
                job->err = "key map entry " + std::to_string(p->mine[g % p->mine.size()]) + " does not point at a start code";
            }
        }
        const auto t = Clock::now();
        if (job->status == LEON_OK) parse_gop(p, job);
        p->st_parse_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t).count();
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->parsed[g] = job;
        }
        p->cv.notify_all();
    }
}


// which of the two parser streams a window's parser kernels run on (LEON_VLC_STREAMS=1: all on one, for A/B runs)
inline size_t vlc_stream_of(int64_t window)
{
    static const bool one = getenv("LEON_VLC_STREAMS") && atoi(getenv("LEON_VLC_STREAMS")) == 1;
    return one ? 0 : (size_t)(window & 1);
}

// gpu_parser: the slices of the whole window in one launch each of k_vlc_parse / k_vlc_index / k_vlc_blocks
// (leon_vlc_gpu.h), on a parser stream, in front of the reconstruction launches that read their output
int launch_gpu_parser(leon_pipeline* p, PipeWindow* w)
{
    hipStream_t vs = p->vlc_stream[vlc_stream_of(w->id)];
    size_t n_slices = 0, n_pics = 0;
    for (GopJob* job : w->jobs) { n_slices += job->slices.size(); n_pics += job->vpics.size(); }
    w->n_vpics = (uint32_t)n_pics;
    if (!n_slices || !n_pics) return LEON_OK;
    VlcRing& R = p->vlc_ring[(size_t)w->ring];
    const size_t desc_bytes = pad256(n_slices * sizeof(leon::VlcSlice)) + pad256(n_pics * sizeof(leon::VlcPic)) + pad256(w->jobs.size() * sizeof(leon::VlcClear));
    const size_t dev_bytes = desc_bytes + pad256(n_slices * sizeof(leon::VlcSliceOut)) + pad256(n_pics * 4);
    if (dev_bytes > R.cap) {                     // the ring entry is ours: its previous window has been released
        if (R.h) hipHostFree(R.h);
        if (R.d) hipFree(R.d);
        R.h = R.d = nullptr;
        R.cap = dev_bytes + dev_bytes / 2;
        if (hipHostMalloc((void**)&R.h, R.cap, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&R.d, R.cap) != hipSuccess) {
            R.cap = 0;
            return fail(LEON_ERR_NOMEM, "descriptor ring of the GPU parser");
        }
    }
    if (n_pics > R.err_cap) {
        if (R.h_err) hipHostFree(R.h_err);
        R.err_cap = n_pics + n_pics / 2;
        if (hipHostMalloc((void**)&R.h_err, R.err_cap * 4, hipHostMallocDefault) != hipSuccess) { R.err_cap = 0; R.h_err = nullptr; return fail(LEON_ERR_NOMEM, "error words of the GPU parser"); }
    }
    leon::VlcSlice* hs = (leon::VlcSlice*)R.h;
    leon::VlcPic* hp = (leon::VlcPic*)(R.h + pad256(n_slices * sizeof(leon::VlcSlice)));
    leon::VlcSliceOut* d_words = (leon::VlcSliceOut*)(R.d + desc_bytes);      // what k_vlc_parse leaves per slice for the two kernels behind it
    uint32_t* d_err = (uint32_t*)(R.d + desc_bytes + pad256(n_slices * sizeof(leon::VlcSliceOut)));
    leon::VlcClear* hc = (leon::VlcClear*)(R.h + pad256(n_slices * sizeof(leon::VlcSlice)) + pad256(n_pics * sizeof(leon::VlcPic)));
    size_t pi = 0, ci = 0;
    // Slice order of the launch: picture by picture, the I pictures first, then P, then B -- the 64 lanes of a wave
    // then hold slices of ONE picture type (in coded order most waves straddled two pictures of different types, and
    // a wave executes the union of its lanes' paths), and the longest waves (I: three times the symbols) start first.
    // The opposite -- the k-th slice of every picture side by side, so that every wave carries a few I slices -- was
    // measured too: 76 k against 97 k pictures/s.
    struct PicRun { const leon::VlcSlice* first; uint32_t n, pic; int type; };
    std::vector<PicRun> runs;
    runs.reserve(n_pics);
    for (GopJob* job : w->jobs) {
        // what the kernels count in and report through starts at zero (regions are multiples of 256 bytes)
        hc[ci].ptr = (uint4*)(job->arena->dev + job->zero_begin);
        hc[ci].n16 = job->zero_bytes / 16;
        ci++;
        size_t at = 0;
        for (size_t k = 0; k < job->vpics.size(); k++) {
            uint32_t n = 0;
            while (at + n < job->slices.size() && job->slices[at + n].pic == k) n++;
            runs.push_back(PicRun{job->slices.data() + at, n, (uint32_t)(pi + k), job->vpics[k].type});
            at += n;
        }
        for (const leon::VlcPic& v : job->vpics) hp[pi++] = v;
    }
    // Inside a type: picture by picture -- a picture's slices stay together (k_vlc_index walks them).  (The k-th slice of
    // every picture side by side, so that the 64 lanes of a wave hold slices that are alike, was measured in round 3:
    // 30 % fewer instructions in all, and 25-30 % SLOWER end to end -- the launch has fewer waves than the chip has
    // room for, it lasts as long as its longest wave, and a wave of 64 long slices is longer than a wave with one.)
    size_t si = 0;
    for (int type = 1; type <= 3; type++)
        for (const PicRun& pr : runs)
            if (pr.type == type) {
                hp[pr.pic].first_slice = (uint32_t)si;
                hp[pr.pic].n_slices = pr.n;
                for (uint32_t k = 0; k < pr.n; k++) {
                    hs[si] = pr.first[k];
                    hs[si].pic = pr.pic;
                    si++;
                }
            }
    HIP_TRY(hipMemcpyAsync(R.d, R.h, desc_bytes, hipMemcpyHostToDevice, vs));
    HIP_TRY(hipMemsetAsync(d_err, 0, n_pics * 4, vs));
    const leon::VlcClear* dc = (const leon::VlcClear*)(R.d + pad256(n_slices * sizeof(leon::VlcSlice)) + pad256(n_pics * sizeof(leon::VlcPic)));
    hipLaunchKernelGGL(leon::k_vlc_clear, dim3(32, (unsigned)w->jobs.size()), dim3(256), 0, vs, dc);
    const leon::VlcSlice* ds = (const leon::VlcSlice*)R.d;
    const leon::VlcPic* dp = (const leon::VlcPic*)(R.d + pad256(n_slices * sizeof(leon::VlcSlice)));
    const int blocks = (int)((n_slices + 255) / 256);
    // LEON_DEBUG_VLC_LDS_PAD (bytes): more dynamic LDS for the slice loop's workgroups than they use -- what the reconstruction beside them loses to the parser's LDS
    static const size_t vlc_lds_pad = getenv("LEON_DEBUG_VLC_LDS_PAD") ? (size_t)atol(getenv("LEON_DEBUG_VLC_LDS_PAD")) : 0;
    hipLaunchKernelGGL(leon::k_vlc_parse, dim3(blocks), dim3(256), 4 * leon::kVlcRingDwords * 64 * 4 + vlc_lds_pad, vs, ds, d_words, (int)n_slices, dp, d_err, p->vgeom, p->d_vlc_tables);
    hipLaunchKernelGGL(leon::k_vlc_index, dim3((unsigned)n_pics), dim3(leon::kVlcIndexThreads), p->vlc_index_lds, vs, ds, d_words, dp, d_err, p->vgeom);
    hipLaunchKernelGGL(leon::k_vlc_blocks, dim3((unsigned)((n_slices + 3) / 4)), dim3(256), 0, vs, ds, d_words, (int)n_slices, dp, d_err, p->vgeom, p->d_vlc_tables);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(R.h_err, d_err, n_pics * 4, hipMemcpyDeviceToHost, vs));
    return LEON_OK;
}

// LEON_DEBUG_CAPTURE=<dir> (read when the pipeline is created; tests/test_pipeline_gpu.py sets it for its small streams):
// what the reconstruction launches of a window READ and WROTE is written to <dir>/w<window>/ -- after every level, with
// the decoder's stream idle, the planes of the slot each picture of the level wrote and of the slots it predicted from
// (L<level>_g<lane>_t<tref>_{out,fwd,bwd}.planes); after the last level every GOP's device arena as the kernels left it
// (arena_<lane>.bin: stream bytes, the GPU parser's records, maps, group offsets, entry lists) with an index of where each
// picture's arrays lie (index.txt).  A wrong frame then comes with the inputs that produced it: which buffer held wrong
// bytes separates a late write into reused pages from a read of memory nobody wrote.  Serialises the levels.
void capture_file(const std::string& path, const void* dev, size_t bytes)
{
    std::vector<char> h(bytes);
    if (bytes && hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(h.data(), 1, bytes, f); fclose(f); }
}

// the pictures of one window as launches: per GOP the anchors rotate through three slots; a picture's
// level is one more than the deepest picture it predicts from, and a level is one batch
int submit_window(leon_pipeline* p, PipeWindow* w)
{
    leon_decoder* d = p->dec;
    const size_t lanes = w->jobs.size();
    struct Item { size_t lane; const PipePic* pic; int fwd, bwd, out; };
    std::vector<std::vector<Item>> levels;
    uint8_t* ring = p->d_rgba + (size_t)w->ring * p->W * p->max_pics * p->frame_bytes;
    w->frames.clear();
    for (size_t j = 0; j < lanes; j++) {
        GopJob* job = w->jobs[j];
        // upload the GOP's arena; everything of this window is copied before its first launch
        const size_t up = p->gpu_parser ? job->upload_bytes : job->arena->used;
        // LEON_DEBUG_POISON=1: everything of the arena that the kernels are expected to write before they read it starts
        // as 0xCD bytes -- a read of something nobody wrote then shows in every run, not only when the memory's history
        // happens to differ from zero
        static const bool poison = getenv("LEON_DEBUG_POISON") && atoi(getenv("LEON_DEBUG_POISON")) == 1;
        if (poison && p->gpu_parser && job->arena->used > up) HIP_TRY(hipMemsetAsync(job->arena->dev + up, 0xCD, job->arena->used - up, p->copy_stream));
        if (up) {
            HIP_TRY(hipMemcpyAsync(job->arena->dev, job->arena->host, up, hipMemcpyHostToDevice, p->copy_stream));
            p->st_upload += up;
        }
        int older = -1, newer = -1, lv_older = -1, lv_newer = -1, n_anchor = 0;      // anchor slots (0..2 of the lane) and their levels
        const int per_lane = 3 + (p->unfused ? p->max_pics : 0);
        int n_b = 0;
        for (const PipePic& m : job->pics) {
            Item it{j, &m, -1, -1, -1};
            int lv = 0;
            if (m.type == LEON_PIC_I) {
                it.out = (int)(per_lane * j) + n_anchor % 3;
            } else if (m.type == LEON_PIC_P) {
                if (newer < 0) return fail(LEON_ERR_INVALID, "GOP %llu: a P picture without a preceding anchor", (unsigned long long)job->gop);
                it.fwd = newer;
                lv = lv_newer + 1;
                it.out = (int)(per_lane * j) + n_anchor % 3;
            } else {
                if (newer < 0) return fail(LEON_ERR_INVALID, "GOP %llu: a B picture without an anchor (open GOPs cannot be sharded)", (unsigned long long)job->key_gop);
                // the leading B pictures of a CLOSED GOP predict backward only (both references = the I picture); in an
                // open GOP (closed_gop = 0) they may predict from the GOP before, which a shard does not have
                if (older < 0 && job->open_gop)
                    return fail(LEON_ERR_INVALID, "GOP %llu is open (closed_gop = 0) and its leading B pictures may predict from the GOP before it: "
                                                  "GOP shards must be closed", (unsigned long long)job->key_gop);
                it.bwd = newer;
                it.fwd = older >= 0 ? older : newer;
                lv = std::max(lv_newer, lv_older) + 1;
                if (p->unfused) it.out = (int)(per_lane * j) + 3 + n_b++ % p->max_pics;      // a slot of its own until it is converted
            }
            if (m.type != LEON_PIC_B) {
                older = newer; lv_older = lv_newer;
                newer = it.out; lv_newer = lv;
                n_anchor++;
            }
            if ((size_t)lv >= levels.size()) levels.resize((size_t)lv + 1);
            levels[(size_t)lv].push_back(it);
            if (m.tref < 0 || m.tref >= p->max_pics) return fail(LEON_ERR_INVALID, "temporal reference %d outside the GOP", m.tref);
        }
        // two pictures of a GOP with one temporal reference would be rendered into the same frame of the ring
        std::vector<uint8_t> seen((size_t)p->max_pics, 0);
        for (const PipePic& m : job->pics) {
            if (seen[(size_t)m.tref]) return fail(LEON_ERR_INVALID, "GOP %llu: two pictures with temporal reference %d", (unsigned long long)job->key_gop, m.tref);
            seen[(size_t)m.tref] = 1;
        }
    }
    hipEvent_t copied = get_event(d);
    if (!copied) return LEON_ERR_HIP;
    HIP_TRY(hipEventRecord(copied, p->copy_stream));
    // LEON_DEBUG_SERIAL=1: the host waits behind every stage (uploads, parser kernels, reconstruction) -- takes every
    // cross-stream dependency out of the picture when a wrong frame is being hunted
    static const bool serial = getenv("LEON_DEBUG_SERIAL") && atoi(getenv("LEON_DEBUG_SERIAL")) == 1;
    if (serial) HIP_TRY(hipStreamSynchronize(p->copy_stream));
    if (p->gpu_parser) {
        // upload -> parser kernels (their own stream) -> reconstruction (the decoder's stream)
        HIP_TRY(hipStreamWaitEvent(p->vlc_stream[vlc_stream_of(w->id)], copied, 0));
        d->ev_pool.push_back(copied);
        const int rc = launch_gpu_parser(p, w);
        if (rc != LEON_OK) return rc;
        if (serial) HIP_TRY(hipStreamSynchronize(p->vlc_stream[vlc_stream_of(w->id)]));
        hipEvent_t parsed = get_event(d);
        if (!parsed) return LEON_ERR_HIP;
        HIP_TRY(hipEventRecord(parsed, p->vlc_stream[vlc_stream_of(w->id)]));
        HIP_TRY(hipStreamWaitEvent(d->stream, parsed, 0));
        d->ev_pool.push_back(parsed);
    } else {
        HIP_TRY(hipStreamWaitEvent(d->stream, copied, 0));
        d->ev_pool.push_back(copied);
    }
    // the matrix sets of shards whose sequence headers carry other matrices than the stream's first: registered with the
    // decoder here, on the one thread that drives it (the upload is ordered in front of the launches by the stream)
    std::vector<std::vector<int32_t>> qset(lanes);
    for (size_t j = 0; j < lanes; j++)
        for (const auto& m : w->jobs[j]->qms) {
            int32_t id = 0;
            const int rc = leon_add_quant_matrices(d, m.data(), m.data() + 64, &id);
            if (rc != LEON_OK) return rc;
            qset[j].push_back(id);
        }
    const bool capture = !p->capture_dir.empty();
    const std::string cdir = capture ? p->capture_dir + "/w" + std::to_string(w->id) : std::string();
    FILE* cidx = nullptr;
    if (capture) {
        (void)mkdir(p->capture_dir.c_str(), 0777);          // (both may exist)
        (void)mkdir(cdir.c_str(), 0777);
        cidx = fopen((cdir + "/index.txt").c_str(), "w");
        if (cidx) fprintf(cidx, "geom coded=%dx%d frame=%dx%d mbs=%d n_groups=%d gpu_parser=%d unfused=%d slot_bytes=%zu\n", p->vinfo.coded_width, p->vinfo.coded_height,
                          p->vinfo.frame_width, p->vinfo.frame_height, p->vinfo.mb_width * p->vinfo.mb_height, p->vinfo.n_groups, (int)p->gpu_parser, (int)p->unfused, d->plane_bytes);
    }
    size_t lvl_no = 0;
    std::vector<leon_sparse_picture> batch;
    for (auto& lvl : levels) {
        batch.clear();
        for (const Item& it : lvl) {
            const char* base = w->jobs[it.lane]->arena->dev;
            const PipePic& m = *it.pic;
            auto ptr = [&](size_t off) -> const void* { return off == kNone ? nullptr : (const void*)(base + off); };
            leon_sparse_picture sp{};
            sp.type = m.type;
            sp.out_slot = it.out;
            sp.ref_fwd_slot = it.fwd;
            sp.ref_bwd_slot = it.bwd;
            sp.grp_off = (const uint32_t*)ptr(m.grp_off);
            sp.entries = (const uint32_t*)ptr(m.entries);
            sp.n_entries = m.n_entries;
            sp.qscale = (const uint8_t*)ptr(m.qscale);
            sp.intra = (const uint8_t*)ptr(m.intra);
            sp.repadd = (const uint8_t*)ptr(m.repadd);
            sp.mb_dir = (const uint8_t*)ptr(m.mb_dir);
            sp.mv_fwd = (const int16_t*)ptr(m.mv_fwd);
            sp.mv_bwd = (const int16_t*)ptr(m.mv_bwd);
            sp.rgba_out = p->unfused ? nullptr : ring + ((size_t)it.lane * p->max_pics + (size_t)m.tref) * p->frame_bytes;
            sp.no_planes = !p->unfused && m.type == LEON_PIC_B;
            sp.qm_set = m.qm >= 0 ? qset[it.lane][(size_t)m.qm] : 0;
            batch.push_back(sp);
        }
        if (batch.empty()) continue;
        int rc = submit_batch_any(d, wrap(batch.data(), (int)batch.size()).data(), (int)batch.size(), LEON_MEM_DEVICE);
        if (rc != LEON_OK) return rc;
        if (p->unfused)
            for (const Item& it : lvl) {
                rc = leon_convert_rgba(d, it.out, ring + ((size_t)it.lane * p->max_pics + (size_t)it.pic->tref) * p->frame_bytes, LEON_MEM_DEVICE, p->flavour);
                if (rc != LEON_OK) return rc;
            }
        if (capture) {
            HIP_TRY(hipStreamSynchronize(d->stream));
            for (const Item& it : lvl) {
                const PipePic& m = *it.pic;
                const std::string stem = cdir + "/L" + std::to_string(lvl_no) + "_g" + std::to_string(it.lane) + "_t" + std::to_string(m.tref);
                if (it.out >= 0) capture_file(stem + "_out.planes", d->d_slots + (size_t)it.out * d->slot_stride, d->plane_bytes);
                if (it.fwd >= 0) capture_file(stem + "_fwd.planes", d->d_slots + (size_t)it.fwd * d->slot_stride, d->plane_bytes);
                if (it.bwd >= 0) capture_file(stem + "_bwd.planes", d->d_slots + (size_t)it.bwd * d->slot_stride, d->plane_bytes);
                if (cidx) fprintf(cidx, "pic lane=%zu gop=%llu tref=%d type=%d level=%zu out=%d fwd=%d bwd=%d n_entries=%u qm=%d grp_off=%zd entries=%zd qscale=%zd intra=%zd repadd=%zd "
                                        "mb_dir=%zd mv_fwd=%zd mv_bwd=%zd\n", it.lane, (unsigned long long)w->jobs[it.lane]->key_gop, m.tref, m.type, lvl_no, it.out, it.fwd, it.bwd,
                                  m.n_entries, m.qm, (ssize_t)m.grp_off, (ssize_t)m.entries, (ssize_t)m.qscale, (ssize_t)m.intra, (ssize_t)m.repadd, (ssize_t)m.mb_dir, (ssize_t)m.mv_fwd, (ssize_t)m.mv_bwd);
            }
        }
        lvl_no++;
    }
    if (capture) {
        for (size_t j = 0; j < lanes; j++) capture_file(cdir + "/arena_" + std::to_string(j) + ".bin", w->jobs[j]->arena->dev, w->jobs[j]->arena->used);
        if (cidx) fclose(cidx);
    }
    if (serial) HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipEventRecord(w->done, d->stream));
    // frames in display order, GOP-major
    const double rate = p->vinfo.picture_rate > 0 ? p->vinfo.picture_rate : 25.0;
    for (size_t j = 0; j < lanes; j++) {
        GopJob* job = w->jobs[j];
        // by temporal reference: a GOP cut short by the encoder may skip display positions
        std::vector<const PipePic*> by_disp((size_t)p->max_pics, nullptr);
        for (const PipePic& m : job->pics) by_disp[(size_t)m.tref] = &m;
        for (size_t k = 0; k < by_disp.size(); k++) {
            if (!by_disp[k]) continue;
            leon_pipeline_frame f{};
            f.gop = job->key_gop;
            f.display_index = (int32_t)k;
            f.type = by_disp[k]->type;
            f.ts_ms = job->gop_ts_ms + 1000.0 * (double)k / rate;
            f.rgba = ring + ((size_t)j * p->max_pics + k) * p->frame_bytes;
            w->frames.push_back(f);
        }
        p->st_pictures += job->pics.size();
    }
    p->st_gops += lanes;
    return LEON_OK;
}

void submit_loop(leon_pipeline* p);
void submit_main(leon_pipeline* p)
{
    hipSetDevice(p->cfg.device_id);
    submit_loop(p);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->submit_exited = true;
    }
    p->cv.notify_all();
}

void submit_loop(leon_pipeline* p)
{
    for (int64_t wid = 0; wid < p->total_windows; wid++) {
        PipeWindow* w = new PipeWindow();
        w->id = wid;
        const auto t_begin = Clock::now();
        auto t_ring = t_begin, t_scanned = t_begin;
        const uint64_t g0 = (uint64_t)wid * p->W, g1 = std::min<uint64_t>(g0 + p->W, p->total_gops);
        {
            std::unique_lock<std::mutex> lk(p->mu);
            // a free entry of the RGBA ring, then every GOP of the window parsed
            p->cv.wait(lk, [&] {
                if (p->stop) return true;
                for (int r = 0; r < p->R; r++)
                    if (p->ring_owner[r] < 0) return true;
                return false;
            });
            if (p->stop) { delete w; return; }
            t_ring = Clock::now();
            for (int r = 0; r < p->R; r++)
                if (p->ring_owner[r] < 0) { w->ring = r; p->ring_owner[r] = wid; break; }
            p->cv.wait(lk, [&] {
                if (p->stop) return true;
                for (uint64_t g = g0; g < g1; g++)
                    if (!p->parsed.count(g)) return false;
                return true;
            });
            if (p->stop) { delete w; return; }
            t_scanned = Clock::now();
            for (uint64_t g = g0; g < g1; g++) {
                w->jobs.push_back(p->parsed[g]);
                p->parsed.erase(g);
            }
        }
        int rc = LEON_OK;
        std::string msg;
        for (GopJob* j : w->jobs)
            if (j->status != LEON_OK && rc == LEON_OK) { rc = j->status; msg = j->err; }
        if (rc == LEON_OK) {
            if (hipEventCreateWithFlags(&w->done, hipEventDisableTiming) != hipSuccess) { rc = LEON_ERR_HIP; msg = "hipEventCreate failed"; }
            else {
                const auto t = Clock::now();
                rc = submit_window(p, w);
                p->st_submit_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t).count();
                if (rc != LEON_OK) msg = g_err;
            }
        }
        w->status = rc;
        {   // where the submit thread's time goes, per window (LEON_PIPE_TRACE=1 prints it when the pipeline is destroyed)
            const auto ns = [](Clock::time_point a, Clock::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
            p->st_wait_ring_ns += ns(t_begin, t_ring);
            p->st_wait_scan_ns += ns(t_ring, t_scanned);
        }
        if (rc != LEON_OK) pipe_fail(p, rc, msg);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->to_notify.push_back(w);
            p->windows_submitted++;
        }
        p->cv.notify_all();
        if (rc != LEON_OK) return;
    }
}

void release_window_locked(leon_pipeline* p, PipeWindow* w)
{
    for (GopJob* j : w->jobs) {
        p->free_arenas.push_back(j->arena);
        delete j;
    }
    p->ring_owner[w->ring] = -1;
    if (w->done) hipEventDestroy(w->done);
    delete w;
}

void notify_main(leon_pipeline* p)
{
    hipSetDevice(p->cfg.device_id);
    for (;;) {
        PipeWindow* w = nullptr;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return !p->to_notify.empty() || p->windows_done == p->total_windows || p->submit_exited; });
            if (p->to_notify.empty()) break;
            w = p->to_notify.front();
            p->to_notify.pop_front();
        }
        if (w->status == LEON_OK && hipEventSynchronize(w->done) != hipSuccess) {
            w->status = LEON_ERR_HIP;
            pipe_fail(p, LEON_ERR_HIP, std::string("window ") + std::to_string(w->id) + ": " + hipGetErrorString(hipGetLastError()));
        }
        if (w->status == LEON_OK && p->gpu_parser && w->n_vpics) {          // the GPU parser's verdict on every picture of the window
            const uint32_t* e = p->vlc_ring[(size_t)w->ring].h_err;
            for (uint32_t k = 0; k < w->n_vpics; k++)
                if (e[k]) {
                    static const char* const what[] = {"", "invalid macroblock address increment", "macroblock address outside the picture",
                        "invalid macroblock type", "invalid motion code", "invalid coded block pattern", "invalid coefficient code",
                        "coefficient index overflow", "bitstream ends inside a slice", "invalid DC size code", "more block records than the picture has blocks",
                                           "slices of a picture overlap (the GPU parser decodes them side by side: gpu_parser = -1 decodes such a stream)"};
                    const uint32_t code = e[k] & 255u;
                    w->status = LEON_ERR_INVALID;
                    pipe_fail(p, LEON_ERR_INVALID, std::string("window ") + std::to_string(w->id) + ", picture " + std::to_string(k) + ", slice " +
                              std::to_string(e[k] >> 8) + ": " + (code < sizeof(what) / sizeof(what[0]) ? what[code] : "malformed slice"));
                    break;
                }
        }
        const int64_t id = w->id;
        const int st = w->status;
        bool quiet;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->delivered[id] = w;
            p->st_seconds = std::chrono::duration<double>(Clock::now() - p->t0).count();
            if (p->st_window_done.size() < 4096) p->st_window_done.push_back(p->st_seconds);
            quiet = p->quiet;
        }
        // frames are handed over outside the lock: the callback may release the window at once
        if (p->cb && !quiet) p->cb(p->user, id, st == LEON_OK ? w->frames.data() : nullptr, st == LEON_OK ? (int32_t)w->frames.size() : 0, st);
        else leon_pipeline_release_window(p, id);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->windows_done++;
        }
        p->cv.notify_all();
    }
    bool quiet;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        quiet = p->quiet;
    }
    // 'ended' (or the error that stopped the run) BEFORE leon_pipeline_wait returns: a host that waits and then
    // looks at what its callback recorded must find the end recorded
    if (p->cb && !quiet) p->cb(p->user, -1, nullptr, 0, p->status);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->finished = true;
    }
    p->cv.notify_all();
}

}  // namespace

extern "C" {

int leon_pipeline_create(const leon_pipeline_config* cfg, const uint8_t* stream, size_t bytes,
                         leon_pipeline_callback cb, void* user, leon_pipeline** out)
{
    return leon_pipeline_create_partial(cfg, stream, bytes, bytes, cb, user, out);
}

int leon_pipeline_feed(leon_pipeline* p, size_t valid_bytes)
{
    if (!p) return fail(LEON_ERR_INVALID, "null pipeline");
    if (valid_bytes > p->bytes) return fail(LEON_ERR_INVALID, "%zu bytes fed, the stream has %zu", valid_bytes, p->bytes);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (valid_bytes > p->valid) p->valid = valid_bytes;
    }
    p->cv.notify_all();
    return LEON_OK;
}

int leon_pipeline_create_partial(const leon_pipeline_config* cfg, const uint8_t* stream, size_t bytes, size_t valid_bytes,
                                 leon_pipeline_callback cb, void* user, leon_pipeline** out)
{
    if (!cfg || !stream || bytes < 16 || !out) return fail(LEON_ERR_INVALID, "null argument");
    if (valid_bytes > bytes) return fail(LEON_ERR_INVALID, "%zu valid bytes of a stream of %zu", valid_bytes, bytes);
    *out = nullptr;
    leon_vlc_stream* st = nullptr;
    // the container header, the key map and the first sequence header must have arrived
    if (leon_vlc_open(stream, valid_bytes, 1, &st) != LEON_VLC_OK) return fail(LEON_ERR_INVALID, "stream: %s", leon_vlc_last_error());
    leon_pipeline* p = new (std::nothrow) leon_pipeline();
    if (!p) { leon_vlc_close(st); return fail(LEON_ERR_NOMEM, "out of host memory"); }
    leon_vlc_get_info(st, &p->vinfo);
    const int n_keys = leon_vlc_get_keymap(st, nullptr, nullptr, 0);
    std::vector<uint32_t> offs((size_t)std::max(n_keys, 0));
    if (n_keys > 0) leon_vlc_get_keymap(st, offs.data(), nullptr, (uint32_t)n_keys);
    uint32_t first_gop = 0;
    if (n_keys > 0 && cfg->start_seconds > 0) {       // the front end's own seek finds the entry (decoders/jsv.js:327-350)
        uint64_t off = 0;
        if (leon_vlc_seek(st, cfg->start_seconds, &off) == LEON_VLC_OK)
            for (int g = 0; g < n_keys; g++)
                if (offs[(size_t)g] == off) first_gop = (uint32_t)g;
    }
    leon_vlc_close(st);
    p->cfg = *cfg;
    p->stream = stream;
    p->bytes = bytes;
    p->valid = valid_bytes;
    p->cb = cb;
    p->user = user;
    if (const char* cd = getenv("LEON_DEBUG_CAPTURE")) p->capture_dir = cd;
    if (n_keys > 0) {
        for (int g = 0; g < n_keys; g++) {
            const uint64_t b = offs[(size_t)g], e = g + 1 < n_keys ? offs[(size_t)g + 1] : bytes;
            // (an entry behind what has arrived of a partial stream is looked at when its bytes are there: parser_main)
            if (b >= e || e > bytes || b + 4 > bytes || (b + 4 <= valid_bytes && (stream[b] != 0 || stream[b + 1] != 0 || stream[b + 2] != 1))) {
                delete p;
                return fail(LEON_ERR_INVALID, "key map entry %d does not point at a start code", g);
            }
            p->shard_begin.push_back(b);
            // the shard takes the start code PREFIX of what follows along (00 00 01, not the code byte): its last slice
            // then ends exactly as it does in the whole stream.  Cut at the key-map offset, a last macroblock coded in
            // two bytes or less would look like the end of the data to the slice loop (jsv.js:1710-1760) and be dropped.
            p->shard_end.push_back(std::min<uint64_t>(e + 3, bytes));
        }
    } else {            // no key map: the whole stream is one shard
        p->shard_begin.push_back(0);
        p->shard_end.push_back(bytes);
    }
    const int loops = cfg->loop > 0 ? cfg->loop : 1;
    const uint32_t sc = cfg->shard_count > 1 ? (uint32_t)cfg->shard_count : 1u, si = sc > 1 ? (uint32_t)cfg->shard_index : 0u;
    if (si >= sc) { delete p; return fail(LEON_ERR_INVALID, "shard_index %d of %d", cfg->shard_index, cfg->shard_count); }
    for (uint32_t g = si; g < p->shard_begin.size(); g += sc)
        if (g >= first_gop) p->mine.push_back(g);
    p->info.first_gop = first_gop;
    if (p->mine.empty()) { delete p; return fail(LEON_ERR_INVALID, "shard %u of %u gets no GOP: the stream has %zu", si, sc, p->shard_begin.size()); }
    p->total_gops = (uint64_t)p->mine.size() * (uint64_t)loops;
    p->W = cfg->gops_per_window > 0 ? cfg->gops_per_window : 32;
    if ((uint64_t)p->W > p->total_gops) p->W = (int)p->total_gops;
    p->gpu_parser = cfg->gpu_parser >= 0;          // 0 = default: the GPU (a pipeline has a device by construction); < 0: the parser threads
    if (cfg->gpu_parser == LEON_PIPELINE_PARSER_DEFAULT) {
        // a caller who did not ASK for the GPU parser is not refused for its limits: a picture whose group counters do not fit
        // k_vlc_index's LDS, or a GOP shard of 2^28 bytes and more (the kernels count bits in 32), goes to the parser threads
        bool fits = ((size_t)p->vinfo.n_groups + leon::kVlcIndexThreads) * 4 <= (size_t)160 * 1024 - 512;
        for (uint32_t g : p->mine) fits = fits && p->shard_end[g] - p->shard_begin[g] < ((uint64_t)1 << 28);
        if (getenv("LEON_DEBUG_GPU_PARSER_LIMIT") && p->vinfo.n_groups > atoi(getenv("LEON_DEBUG_GPU_PARSER_LIMIT"))) fits = false;      // tests: a mocked n_groups limit
        if (!fits) p->gpu_parser = false;
    }
    p->info.gpu_parser = p->gpu_parser ? 1 : 0;
    p->R = cfg->windows_in_flight > 0 ? cfg->windows_in_flight : (p->gpu_parser ? 3 : 2);      // GPU parser: one window being parsed beside one reconstructed and one read
    if (cfg->max_gop_pictures > 0) p->max_pics = cfg->max_gop_pictures;
    else if (valid_bytes < bytes) p->max_pics = 16;          // still arriving: nothing to count yet
    else {
        // the frames a GOP can fill in a window's ring entry: the longest GOP of this pipeline's shards, from the picture
        // start codes (00 00 01 00 -- the syntax keeps that pattern out of everything else); a fixed 16 would reserve a
        // third more than IBBP-12 streams use
        size_t longest = 1;
        for (uint64_t g : p->mine) {
            const uint8_t* b = stream + p->shard_begin[g];
            const uint8_t* const e = stream + p->shard_end[g];
            size_t n = 0;
            while (b + 4 <= e) {
                const uint8_t* z = (const uint8_t*)memchr(b, 0, (size_t)(e - b) - 3);
                if (!z) break;
                if (z[1] == 0 && z[2] == 1 && z[3] == 0) { n++; b = z + 4; } else b = z + 1;
            }
            longest = std::max(longest, n);
        }
        p->max_pics = (int)std::min<size_t>(longest, 1024);
    }
    int k = cfg->parser_threads;
    if (k <= 0) { k = (int)std::thread::hardware_concurrency(); if (k < 1) k = 1; if (k > 16) k = 16; }
    p->K = k;
    p->total_windows = (int64_t)((p->total_gops + p->W - 1) / p->W);
    p->frame_bytes = (size_t)p->vinfo.frame_width * p->vinfo.frame_height * 4;
    p->info.coded_width = p->vinfo.coded_width; p->info.coded_height = p->vinfo.coded_height;
    p->info.frame_width = p->vinfo.frame_width; p->info.frame_height = p->vinfo.frame_height;
    p->info.picture_rate = p->vinfo.picture_rate; p->info.duration = p->vinfo.duration;
    p->info.gops = (uint32_t)p->shard_begin.size();
    p->info.shard_gops = (uint32_t)p->mine.size();
    p->info.parser_threads = p->K; p->info.gops_per_window = p->W;
    // the fused display conversion writes eight pixels per lane: it needs frame_width % 8 == 0.  The reference crops to
    // any width (player/easybits.player.js:2818): such a stream takes the slow road -- every picture (B pictures too)
    // writes its planes, and one leon_convert_rgba per picture (the generic k_rgba_twin) fills the window's frames.
    // The GL flavour (the reference's live display arithmetic, fp32) exists as a launch of its own only: the same road.
    if (cfg->display_flavour != LEON_RGB_CPU_TWIN && cfg->display_flavour != LEON_RGB_GL) { delete p; return fail(LEON_ERR_INVALID, "display_flavour %d", cfg->display_flavour); }
    p->flavour = cfg->display_flavour;
    p->info.display_flavour = p->flavour;
    p->unfused = (p->vinfo.frame_width & 7) != 0 || p->flavour == LEON_RGB_GL;
    if (p->unfused && p->vinfo.has_alpha == 1) {
        delete p;
        return fail(LEON_ERR_INVALID, "a yuva stream needs frame_width %% 8 == 0 (it is %d) and the CPU-twin display flavour in the pipeline", p->vinfo.frame_width);
    }

    leon_config dc{};
    dc.coded_width = p->vinfo.coded_width; dc.coded_height = p->vinfo.coded_height;
    dc.frame_width = p->vinfo.frame_width; dc.frame_height = p->vinfo.frame_height;
    dc.n_slots = (3 + (p->unfused ? p->max_pics : 0)) * p->W;      // three rotating anchors per GOP of a window (+ its B pictures)
    dc.alpha = p->vinfo.has_alpha == 1;        // yuva: the frames' A bytes come from the stream's fourth component
    dc.device_id = cfg->device_id;
    int rc = leon_create(&dc, &p->dec);
    if (rc != LEON_OK) { delete p; return rc; }
    rc = leon_set_quant_matrices(p->dec, p->vinfo.intra_qm, p->vinfo.non_intra_qm);
    auto bail = [&](int code, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(hipGetLastError());
        leon_pipeline_destroy(p);
        return fail(code, "%s", m.c_str());
    };
    if (rc != LEON_OK) { leon_pipeline_destroy(p); return rc; }
    if (hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) != hipSuccess) return bail(LEON_ERR_HIP, "copy stream");
    if (big_alloc((void**)&p->d_rgba, (size_t)p->R * p->W * p->max_pics * p->frame_bytes, kBigRgbaRing) != hipSuccess) {
        const std::string what = "RGBA ring of " + std::to_string((size_t)p->R * p->W * p->max_pics * p->frame_bytes >> 20) + " MiB (windows_in_flight " +
                                 std::to_string(p->R) + " x gops_per_window " + std::to_string(p->W) + " x max_gop_pictures " + std::to_string(p->max_pics) +
                                 " x " + std::to_string(p->frame_bytes) + " bytes per frame)";
        return bail(LEON_ERR_NOMEM, what.c_str());
    }
    p->ring_owner.assign((size_t)p->R, -1);
    if (p->gpu_parser) {
        // the front end's tables in the order the kernels copy them to LDS (leon_vlc_gpu.h)
        std::vector<leon_vlc_gpu_tables> src(1);
        std::vector<leon::VlcTables> t(1);
        leon_vlc_get_gpu_tables(src.data());
        // 16 bits per entry for the part that lives in LDS: fast12 as {length:4, end of block:1, run:5, level:6},
        // the others as (length << 8) | value
        for (int i = 0; i < 4096; i++) {
            const uint32_t f = src[0].fast12[i];
            const int len = (int)(f & 0x7f), run = (int)((f >> 8) & 0xff), level = (int)(int16_t)(f >> 16);
            if (len > 15 || run > 31 || level < -32 || level > 31) { leon_pipeline_destroy(p); return fail(LEON_ERR_INVALID, "coefficient table does not fit 16 bits"); }
            t[0].fast12[i] = (uint16_t)(len | ((f & 0x80u) ? 0x10 : 0) | (run << 5) | ((level & 63) << 10));
        }
        // multi12: every symbol that lies complete in the next 12 bits, taken together (a prefix code is decided by its own
        // bits: the entry of the pattern shifted up, whatever follows, names the symbol if its length fits what is left)
        for (int i = 0; i < leon::kVlcMulti; i++) {
            const int B = leon::kVlcMultiBits;
            int pos = 0, nsym = 0, adv = 0, eob = 0;
            while (pos < B) {
                // the 12 bits from `pos` on, zeros behind the pattern's end
                const uint32_t f = src[0].fast12[(((uint32_t)i << pos) & (uint32_t)(leon::kVlcMulti - 1)) << 12 >> B];
                const int len = (int)(f & 0x7f);
                if (len == 0 || len > B - pos) break;
                pos += len;
                if (f & 0x80u) { eob = 1; break; }
                nsym++;
                adv += (int)((f >> 8) & 0xff) + 1;
            }
            if (pos > 15 || nsym > 7 || adv > 255) { leon_pipeline_destroy(p); return fail(LEON_ERR_INVALID, "multi-symbol table does not fit 16 bits"); }
            t[0].multi12[i] = (uint16_t)(pos | (nsym << 4) | (eob << 7) | (adv << 8));
        }
        auto pack = [](int32_t e) { return (uint16_t)(((e >> 16) << 8) | (e & 0xff)); };
        for (int i = 0; i < 2048; i++) { t[0].motion_s[i] = pack(src[0].motion_s[i]); t[0].mba[i] = pack(src[0].mba[i]); }
        for (int i = 0; i < 512; i++) t[0].cbp[i] = pack(src[0].cbp[i]);
        for (int k = 0; k < 4; k++) for (int i = 0; i < 64; i++) t[0].mbtype[k][i] = pack(src[0].mbtype[k][i]);
        for (int i = 0; i < 128; i++) t[0].dc_lum[i] = pack(src[0].dc_lum[i]);
        for (int i = 0; i < 256; i++) t[0].dc_chr[i] = pack(src[0].dc_chr[i]);
        for (int i = 0; i < 64; i++) t[0].zz_off[i] = src[0].zz_off[i];
        // the codes of 12 .. 16 bits start with seven zeros: by the nine bits behind them (= the first 512 entries of the
        // 16-bit table)
        for (int i = 0; i < 512; i++) {
            const int32_t e = src[0].coef16[i];
            const int len = e >> 16, cf = e & 0xffff, run = cf >> 8, level = cf & 0xff;
            if (e == 0) { t[0].long9[i] = 0; continue; }
            if (cf == 0xffff || len < 8 || len > 16 || run > 31 || level > 63 || level == 0) { leon_pipeline_destroy(p); return fail(LEON_ERR_INVALID, "long coefficient code does not fit 16 bits"); }
            t[0].long9[i] = (uint16_t)(len | (run << 5) | (level << 10));
        }
        // The parser kernels want little of the chip (a quarter of the issue slots of the SIMDs they sit on) but all of it
        // for as long as their longest slice takes; the reconstruction launches beside them fill every CU.  Highest
        // stream priority: a freed slot goes to a waiting parser workgroup first.  (LEON_VLC_PRIO=0: default priority.)
        int prio_lo = 0, prio_hi = 0;
        hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        const bool high = !(getenv("LEON_VLC_PRIO") && atoi(getenv("LEON_VLC_PRIO")) == 0);
        // LEON_VLC_CUS=K: the parser streams may use K compute units only (hipExtStreamCreateWithCUMask; the first K bits of
        // the mask).  Packed four waves to a SIMD on a part of the chip, the parser leaves the rest to the reconstruction
        // launches at their full occupancy instead of thinning them out everywhere.
        const int cus = getenv("LEON_VLC_CUS") ? atoi(getenv("LEON_VLC_CUS")) : 0;
        for (hipStream_t& vs : p->vlc_stream) {
            hipError_t e;
            if (cus > 0) {
                uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < cus && i < 256; i++) mask[i >> 5] |= 1u << (i & 31);
                e = hipExtStreamCreateWithCUMask(&vs, 8, mask);
            } else e = hipStreamCreateWithPriority(&vs, hipStreamNonBlocking, high ? prio_hi : prio_lo);
            if (e != hipSuccess) return bail(LEON_ERR_HIP, "parser stream");
        }
        if (hipMalloc((void**)&p->d_vlc_tables, sizeof(leon::VlcTables)) != hipSuccess) return bail(LEON_ERR_NOMEM, "GPU parser tables");
        if (hipMemcpy(p->d_vlc_tables, t.data(), sizeof(leon::VlcTables), hipMemcpyHostToDevice) != hipSuccess) return bail(LEON_ERR_HIP, "GPU parser tables");
        p->vgeom.mbw = p->vinfo.mb_width; p->vgeom.mbh = p->vinfo.mb_height;
        p->vgeom.gy = p->vinfo.groups_y; p->vgeom.gc = p->vinfo.groups_c;
        p->vgeom.n_y = 2 * p->vinfo.mb_height * p->vinfo.groups_y;
        p->vgeom.n_c = p->vinfo.mb_height * p->vinfo.groups_c;
        p->vgeom.n_groups = p->vinfo.n_groups;
        p->vgeom.alpha = p->vinfo.has_alpha == 1;
        {   // a picture's maps, in the order scan_gop_for_gpu lays them out
            const size_t mbs = (size_t)p->vinfo.mb_width * p->vinfo.mb_height;
            const size_t mpad = pad256(mbs), vpad = pad256(mbs * 4);
            p->vgeom.off_qscale = 0;
            p->vgeom.off_intra = (uint32_t)mpad;
            p->vgeom.off_repadd = (uint32_t)(2 * mpad);
            p->vgeom.off_mb_dir = (uint32_t)(3 * mpad);
            p->vgeom.off_mv_fwd = (uint32_t)(4 * mpad);
            p->vgeom.off_mv_bwd = (uint32_t)(4 * mpad + vpad);
        }
        // k_vlc_index counts a picture's groups in LDS (one workgroup may have all 160 KiB of a CU)
        p->vlc_index_lds = ((size_t)p->vinfo.n_groups + leon::kVlcIndexThreads) * 4;
        if (p->vlc_index_lds > 160 * 1024 - 512) return bail(LEON_ERR_INVALID, "picture too large for the GPU parser (its group counters do not fit in LDS): use LEON_PIPELINE_PARSER_HOST or LEON_PIPELINE_PARSER_DEFAULT");
        if (p->vlc_index_lds > 48 * 1024 &&
            hipFuncSetAttribute((const void*)leon::k_vlc_index, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->vlc_index_lds) != hipSuccess)
            return bail(LEON_ERR_HIP, "LDS of the GPU parser's index kernel");
        p->vlc_ring.resize((size_t)p->R);
    }
    const int n_arenas = p->W * (p->R + 1);
    for (int i = 0; i < n_arenas; i++) {
        Arena* a = new Arena();
        p->all_arenas.push_back(a);
        p->free_arenas.push_back(a);
    }
    if (p->gpu_parser) {
        // The arenas of the GPU parser come out of two slabs allocated HERE, sized for the largest GOP shard of the key
        // map (what scan_gop_for_gpu will ask for, from its own bounds: 32 / 3 bytes of entry list per stream byte -- an entry
        // takes three bits at least --, the maps, macroblock records and block records of max_gop_pictures pictures; round 3
        // asked for 48 bytes per stream byte, 35 GB at W = 128 in 1080p, round 4's bounds come to 15 GB): allocated one by one at
        // their first use -- and again when a larger GOP came by, hipFree waits for the device -- the first four windows of
        // a 1080p run took 35-40 ms each instead of 8.6.  A GOP that still does not fit gets an allocation of its own
        // (arena_reserve); without the memory for the slabs everything does.
        size_t largest = 0;
        for (uint64_t g : p->mine) largest = std::max(largest, (size_t)(p->shard_end[g] - p->shard_begin[g]));
        const size_t mbs = (size_t)p->vinfo.mb_width * p->vinfo.mb_height;
        const size_t per_pic = pad256(mbs * leon::kVlcMbRecBytes) + 4 * pad256(mbs) + 2 * pad256(mbs * 4) + pad256(((size_t)p->vinfo.n_groups + 1) * 4) + 1024 +
                               pad256(mbs * (size_t)leon::vlc_blocks_per_mb(p->vinfo.has_alpha == 1) * leon::kVlcRecWords * 4) +
                               8 * (size_t)p->vinfo.mb_height * 4;      // (+ 2 entries per slice: one slice per macroblock row is the common case, more find room in the lists' slack)
        const size_t host_each = pad256(largest + 16) + 4096;
        const size_t dev_each = (pad256(largest + 16) + 11 * largest + (size_t)p->max_pics * per_pic + 65535) / 65536 * 65536;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)n_arenas * dev_each < free_b / 2 && !getenv("LEON_DEBUG_NO_SLABS")) {
            if (big_alloc((void**)&p->slab_dev, (size_t)n_arenas * dev_each, kBigArenas) == hipSuccess &&
                hipHostMalloc((void**)&p->slab_host, (size_t)n_arenas * host_each, hipHostMallocDefault) == hipSuccess) {
                for (int i = 0; i < n_arenas; i++) {
                    Arena* a = p->all_arenas[(size_t)i];
                    a->dev = p->slab_dev + (size_t)i * dev_each; a->cap = dev_each; a->dev_owned = false;
                    a->host = p->slab_host + (size_t)i * host_each; a->host_cap = host_each; a->host_owned = false;
                }
            } else {
                if (p->slab_dev) big_free(p->slab_dev);
                p->slab_dev = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    p->t0 = Clock::now();
    for (int i = 0; i < p->K; i++) p->parsers.emplace_back(parser_main, p);
    p->submitter = std::thread(submit_main, p);
    p->notifier = std::thread(notify_main, p);
    *out = p;
    return LEON_OK;
}

int leon_pipeline_get_info(leon_pipeline* p, leon_pipeline_info* out)
{
    if (!p || !out) return fail(LEON_ERR_INVALID, "null argument");
    *out = p->info;
    return LEON_OK;
}

int leon_pipeline_release_window(leon_pipeline* p, int64_t window)
{
    if (!p) return fail(LEON_ERR_INVALID, "null pipeline");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        auto it = p->delivered.find(window);
        if (it == p->delivered.end()) return fail(LEON_ERR_INVALID, "window %lld is not out for delivery", (long long)window);
        release_window_locked(p, it->second);
        p->delivered.erase(it);
    }
    p->cv.notify_all();
    return LEON_OK;
}

int leon_pipeline_wait(leon_pipeline* p)
{
    if (!p) return fail(LEON_ERR_INVALID, "null pipeline");
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->finished; });
    if (p->status != LEON_OK) return fail(p->status, "%s", p->err.c_str());
    return LEON_OK;
}

int leon_pipeline_get_stats(leon_pipeline* p, leon_pipeline_stats* out)
{
    if (!p || !out) return fail(LEON_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    out->pictures = p->st_pictures;
    out->gops = p->st_gops;
    out->windows = (uint64_t)p->windows_done;
    out->stream_bytes = p->bytes;
    out->seconds = p->st_seconds;
    out->parse_seconds_sum = (double)p->st_parse_ns.load() * 1e-9;
    out->upload_bytes = (double)p->st_upload.load();
    out->entries = p->st_entries;
    return LEON_OK;
}

int leon_pipeline_read_frame(leon_pipeline* p, const leon_pipeline_frame* f, uint8_t* rgba_host)
{
    if (!p || !f || !f->rgba || !rgba_host) return fail(LEON_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(p->cfg.device_id));
    HIP_TRY(hipMemcpy(rgba_host, f->rgba, p->frame_bytes, hipMemcpyDeviceToHost));
    return LEON_OK;
}

const char* leon_pipeline_error(leon_pipeline* p)
{
    if (!p) return "";
    std::lock_guard<std::mutex> lk(p->mu);
    return p->err.c_str();
}

void leon_pipeline_destroy(leon_pipeline* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
        p->quiet = true;           // no callbacks while tearing down
    }
    p->cv.notify_all();
    for (auto& t : p->parsers) if (t.joinable()) t.join();
    if (p->submitter.joinable()) p->submitter.join();
    if (p->notifier.joinable()) p->notifier.join();      // drains what was submitted, releasing instead of delivering
    hipSetDevice(p->cfg.device_id);
    // every stream of the pipeline is idle before any of its memory is freed (a submit interrupted by `stop` may have
    // left uploads or parser kernels behind that no window's event covers)
    if (p->copy_stream) hipStreamSynchronize(p->copy_stream);
    for (hipStream_t vs : p->vlc_stream)
        if (vs) hipStreamSynchronize(vs);
    if (p->dec) leon_sync(p->dec);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        for (auto& kv : p->delivered) release_window_locked(p, kv.second);
        p->delivered.clear();
        for (auto& kv : p->parsed) delete kv.second;
        p->parsed.clear();
    }
    for (Arena* a : p->all_arenas) {
        if (a->host && a->host_owned) hipHostFree(a->host);
        if (a->dev && a->dev_owned) big_free(a->dev);
        delete a;
    }
    if (p->slab_host) hipHostFree(p->slab_host);
    if (p->slab_dev) big_free(p->slab_dev);
    if (getenv("LEON_DEBUG_PIPE_TIMING"))      // where the submit thread's time went (LEON_DEBUG_PIPE_TIMING=1)
        fprintf(stderr, "leon pipeline: %llu windows; per window on the submit thread: %.2f ms waiting for a ring entry, %.2f ms waiting for the window's GOPs "
                        "(parser threads), %.2f ms in submit_window\n", (unsigned long long)p->windows_submitted,
                p->windows_submitted ? (double)p->st_wait_ring_ns / 1e6 / (double)p->windows_submitted : 0.0,
                p->windows_submitted ? (double)p->st_wait_scan_ns / 1e6 / (double)p->windows_submitted : 0.0,
                p->windows_submitted ? (double)p->st_submit_ns.load() / 1e6 / (double)p->windows_submitted : 0.0);
    if (getenv("LEON_DEBUG_PIPE_TIMING") && !p->st_window_done.empty()) {
        fprintf(stderr, "leon pipeline: windows completed at (ms):");
        for (size_t i = 0; i < p->st_window_done.size() && i < 64; i++) fprintf(stderr, " %.1f", p->st_window_done[i] * 1e3);
        fprintf(stderr, "\n");
    }
    for (VlcRing& r : p->vlc_ring) {
        if (r.h) hipHostFree(r.h);
        if (r.d) hipFree(r.d);
        if (r.h_err) hipHostFree(r.h_err);
    }
    if (p->d_vlc_tables) hipFree(p->d_vlc_tables);
    for (hipStream_t vs : p->vlc_stream)
        if (vs) hipStreamDestroy(vs);
    if (p->d_rgba) big_free(p->d_rgba);
    if (p->copy_stream) hipStreamDestroy(p->copy_stream);
    if (p->dec) leon_destroy(p->dec);
    delete p;
}

}  // extern "C"
