// leon_napi.cc -- thin N-API addon: JavaScript (Node) -> the C ABI of include/leon.h.
// Nothing is computed here; every function forwards to libleon_hip.so and throws a
// JavaScript Error carrying leon_last_error() on failure (the reference throws from
// its GL layer the same way: decoders/jsv.js:120-122, :231-233, :1175).
//
//   const leon = require('./leon_napi.node');
//   const h = leon.create({codedWidth, codedHeight, frameWidth, frameHeight, nSlots, deviceId});
//   h.setQuantMatrices(u8[64], u8[64]); h.acquireSlot(); h.releaseSlot(s); h.freeDecodedSlots();
//   h.submitPicture({type, outSlot, refFwdSlot, refBwdSlot, coefY, coefCb, coefCr, qscale, intra,
//                    repadd, mvFwd, mvBwd, mbDir});            // = jsv.prototype.IDCT_GL
//   h.submitSparse({type, outSlot, refFwdSlot, refBwdSlot, grpOff, entries, nEntries, qscale, intra,
//                   repadd, mvFwd, mvBwd, mbDir});             // the same through the sparse boundary
//   h.convertRGBA(slot, flavour) -> Uint8Array; h.readPlanes(slot) -> {y, cb, cr}; h.sync(); h.destroy();
#include <node_api.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/leon.h"
#include "../../include/leon_pipeline.h"
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Handle {
    leon_decoder* d;
    leon_config cfg;
};

#define NAPI_OK(call)                                                      \
    do {                                                                   \
        if ((call) != napi_ok) {                                           \
            napi_throw_error(env, nullptr, "N-API call failed: " #call);   \
            return nullptr;                                                \
        }                                                                  \
    } while (0)

napi_value throw_leon(napi_env env, int rc)
{
    char msg[600];
    snprintf(msg, sizeof msg, "leon error %d: %s", rc, leon_last_error());
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

bool get_i32(napi_env env, napi_value obj, const char* key, int32_t* out, int32_t dflt)
{
    napi_value v;
    bool has = false;
    *out = dflt;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return true;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok) return false;
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (t == napi_undefined || t == napi_null) return true;
    return napi_get_value_int32(env, v, out) == napi_ok;
}

// typed array property -> pointer (nullptr when absent/null); checks the element count
bool get_array(napi_env env, napi_value obj, const char* key, napi_typedarray_type want, size_t min_len, const void** out)
{
    *out = nullptr;
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return true;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok) return false;
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (t == napi_undefined || t == napi_null) return true;
    bool is_ta = false;
    napi_is_typedarray(env, v, &is_ta);
    if (!is_ta) return false;
    napi_typedarray_type type;
    size_t len;
    void* data;
    if (napi_get_typedarray_info(env, v, &type, &len, &data, nullptr, nullptr) != napi_ok) return false;
    if (type != want || len < min_len) return false;
    *out = data;
    return true;
}

Handle* unwrap(napi_env env, napi_callback_info info, size_t* argc, napi_value* argv)
{
    napi_value self;
    if (napi_get_cb_info(env, info, argc, argv, &self, nullptr) != napi_ok) return nullptr;
    Handle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) != napi_ok || !h || !h->d) {
        napi_throw_error(env, nullptr, "leon: decoder handle is destroyed or invalid");
        return nullptr;
    }
    return h;
}

void finalize(napi_env, void* data, void*)
{
    Handle* h = (Handle*)data;
    if (h->d) leon_destroy(h->d);
    delete h;
}

napi_value SetQuantMatrices(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    const uint8_t* m[2] = {nullptr, nullptr};
    for (size_t i = 0; i < argc && i < 2; i++) {
        napi_valuetype t;
        napi_typeof(env, argv[i], &t);
        if (t == napi_undefined || t == napi_null) continue;
        napi_typedarray_type type;
        size_t len;
        void* data;
        if (napi_get_typedarray_info(env, argv[i], &type, &len, &data, nullptr, nullptr) != napi_ok || type != napi_uint8_array || len < 64) {
            napi_throw_type_error(env, nullptr, "setQuantMatrices: Uint8Array(64) expected");
            return nullptr;
        }
        m[i] = (const uint8_t*)data;
    }
    int rc = leon_set_quant_matrices(h->d, m[0], m[1]);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value AcquireSlot(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    Handle* h = unwrap(env, info, &argc, nullptr);
    if (!h) return nullptr;
    int32_t slot = -1;
    int rc = leon_acquire_slot(h->d, &slot);
    if (rc != LEON_OK) return throw_leon(env, rc);          // "no free render buffers"
    napi_value v;
    NAPI_OK(napi_create_int32(env, slot, &v));
    return v;
}

napi_value ReleaseSlot(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    int32_t slot = -1;
    if (argc < 1 || napi_get_value_int32(env, argv[0], &slot) != napi_ok) {
        napi_throw_type_error(env, nullptr, "releaseSlot(slot)");
        return nullptr;
    }
    int rc = leon_release_slot(h->d, slot);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value FreeDecodedSlots(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    Handle* h = unwrap(env, info, &argc, nullptr);
    if (!h) return nullptr;
    int rc = leon_free_decoded_slots(h->d);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value SubmitPicture(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    if (argc < 1) {
        napi_throw_type_error(env, nullptr, "submitPicture(picture)");
        return nullptr;
    }
    napi_value p = argv[0];
    leon_picture pic;
    memset(&pic, 0, sizeof pic);
    size_t ny = (size_t)h->cfg.coded_width * h->cfg.coded_height, nc = ny / 4;
    size_t mbs = (size_t)(h->cfg.coded_width / 16) * (h->cfg.coded_height / 16);
    bool ok = get_i32(env, p, "type", &pic.type, 0) && get_i32(env, p, "outSlot", &pic.out_slot, -1) &&
              get_i32(env, p, "refFwdSlot", &pic.ref_fwd_slot, -1) && get_i32(env, p, "refBwdSlot", &pic.ref_bwd_slot, -1) &&
              get_array(env, p, "coefY", napi_int16_array, ny, (const void**)&pic.coef_y) &&
              get_array(env, p, "coefCb", napi_int16_array, nc, (const void**)&pic.coef_cb) &&
              get_array(env, p, "coefCr", napi_int16_array, nc, (const void**)&pic.coef_cr) &&
              get_array(env, p, "coefA", napi_int16_array, ny, (const void**)&pic.coef_a) &&
              get_array(env, p, "qscale", napi_uint8_array, mbs, (const void**)&pic.qscale) &&
              get_array(env, p, "intra", napi_uint8_array, mbs, (const void**)&pic.intra) &&
              get_array(env, p, "repadd", napi_uint8_array, mbs, (const void**)&pic.repadd) &&
              get_array(env, p, "mvFwd", napi_int16_array, mbs * 2, (const void**)&pic.mv_fwd) &&
              get_array(env, p, "mvBwd", napi_int16_array, mbs * 2, (const void**)&pic.mv_bwd) &&
              get_array(env, p, "mbDir", napi_uint8_array, mbs, (const void**)&pic.mb_dir);
    if (!ok) {
        napi_throw_type_error(env, nullptr, "submitPicture: a boundary tensor has the wrong type or is too short");
        return nullptr;
    }
    int rc = leon_submit_picture(h->d, &pic);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

// = IDCT_GL with the coefficients as sparse group lists (include/leon_vlc.h): what
// leon_vlc_napi's nextPicture() returns goes in unchanged
napi_value SubmitSparse(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    if (argc < 1) {
        napi_throw_type_error(env, nullptr, "submitSparse(picture)");
        return nullptr;
    }
    napi_value p = argv[0];
    leon_sparse_picture pic;
    memset(&pic, 0, sizeof pic);
    const int mbw = h->cfg.coded_width / 16, mbh = h->cfg.coded_height / 16;
    const size_t mbs = (size_t)mbw * mbh;
    const size_t n_groups = (size_t)2 * mbh * ((2 * mbw + 7) / 8) * (h->cfg.alpha ? 2 : 1) + (size_t)2 * mbh * ((mbw + 7) / 8);
    int32_t n_entries = 0;
    bool ok = get_i32(env, p, "type", &pic.type, 0) && get_i32(env, p, "outSlot", &pic.out_slot, -1) &&
              get_i32(env, p, "refFwdSlot", &pic.ref_fwd_slot, -1) && get_i32(env, p, "refBwdSlot", &pic.ref_bwd_slot, -1) &&
              get_i32(env, p, "nEntries", &n_entries, 0) && n_entries >= 0 &&
              get_array(env, p, "grpOff", napi_uint32_array, n_groups + 1, (const void**)&pic.grp_off) &&
              get_array(env, p, "entries", napi_uint32_array, (size_t)n_entries, (const void**)&pic.entries) &&
              get_array(env, p, "qscale", napi_uint8_array, mbs, (const void**)&pic.qscale) &&
              get_array(env, p, "intra", napi_uint8_array, mbs, (const void**)&pic.intra) &&
              get_array(env, p, "repadd", napi_uint8_array, mbs, (const void**)&pic.repadd) &&
              get_array(env, p, "mvFwd", napi_int16_array, mbs * 2, (const void**)&pic.mv_fwd) &&
              get_array(env, p, "mvBwd", napi_int16_array, mbs * 2, (const void**)&pic.mv_bwd) &&
              get_array(env, p, "mbDir", napi_uint8_array, mbs, (const void**)&pic.mb_dir);
    if (!ok) {
        napi_throw_type_error(env, nullptr, "submitSparse: a boundary tensor has the wrong type or is too short");
        return nullptr;
    }
    pic.n_entries = (uint32_t)n_entries;
    int rc = leon_submit_sparse(h->d, &pic, 1, LEON_MEM_HOST);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value make_u8(napi_env env, size_t bytes, uint8_t** data)
{
    napi_value ab, ta;
    if (napi_create_arraybuffer(env, bytes, (void**)data, &ab) != napi_ok) return nullptr;
    if (napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta) != napi_ok) return nullptr;
    return ta;
}

napi_value ConvertRGBA(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    int32_t slot = -1, flavour = 0;
    if (argc < 1 || napi_get_value_int32(env, argv[0], &slot) != napi_ok) {
        napi_throw_type_error(env, nullptr, "convertRGBA(slot[, flavour])");
        return nullptr;
    }
    if (argc > 1) napi_get_value_int32(env, argv[1], &flavour);
    uint8_t* data = nullptr;
    napi_value ta = make_u8(env, (size_t)h->cfg.frame_width * h->cfg.frame_height * 4, &data);
    if (!ta) return nullptr;
    int rc = leon_convert_rgba(h->d, slot, data, LEON_MEM_HOST, flavour);
    return rc == LEON_OK ? ta : throw_leon(env, rc);
}

napi_value ReadPlanes(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    int32_t slot = -1;
    if (argc < 1 || napi_get_value_int32(env, argv[0], &slot) != napi_ok) {
        napi_throw_type_error(env, nullptr, "readPlanes(slot)");
        return nullptr;
    }
    size_t ny = (size_t)h->cfg.coded_width * h->cfg.coded_height, nc = ny / 4;
    uint8_t *y, *cb, *cr;
    napi_value ty = make_u8(env, ny, &y), tcb = make_u8(env, nc, &cb), tcr = make_u8(env, nc, &cr);
    if (!ty || !tcb || !tcr) return nullptr;
    int rc = leon_read_planes(h->d, slot, y, cb, cr);
    if (rc != LEON_OK) return throw_leon(env, rc);
    napi_value o;
    NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "y", ty);
    napi_set_named_property(env, o, "cb", tcb);
    napi_set_named_property(env, o, "cr", tcr);
    if (h->cfg.alpha) {                       // yuva: the fourth plane of the slot
        uint8_t* a;
        napi_value ta = make_u8(env, ny, &a);
        if (!ta) return nullptr;
        rc = leon_read_alpha_plane(h->d, slot, a);
        if (rc != LEON_OK) return throw_leon(env, rc);
        napi_set_named_property(env, o, "a", ta);
    }
    return o;
}

napi_value Sync(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    Handle* h = unwrap(env, info, &argc, nullptr);
    if (!h) return nullptr;
    int rc = leon_sync(h->d);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value Destroy(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    napi_value self;
    napi_get_cb_info(env, info, &argc, nullptr, &self, nullptr);
    Handle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) == napi_ok && h && h->d) {
        leon_destroy(h->d);
        h->d = nullptr;
    }
    return nullptr;
}

napi_value Create(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 1) {
        napi_throw_type_error(env, nullptr, "create({codedWidth, codedHeight, ...})");
        return nullptr;
    }
    Handle* h = new Handle();
    memset(&h->cfg, 0, sizeof h->cfg);
    h->d = nullptr;
    bool ok = get_i32(env, argv[0], "codedWidth", &h->cfg.coded_width, 0) && get_i32(env, argv[0], "codedHeight", &h->cfg.coded_height, 0) &&
              get_i32(env, argv[0], "frameWidth", &h->cfg.frame_width, 0) && get_i32(env, argv[0], "frameHeight", &h->cfg.frame_height, 0) &&
              get_i32(env, argv[0], "nSlots", &h->cfg.n_slots, 13) && get_i32(env, argv[0], "deviceId", &h->cfg.device_id, 0) &&
              get_i32(env, argv[0], "alpha", &h->cfg.alpha, 0);
    if (!ok) {
        delete h;
        napi_throw_type_error(env, nullptr, "create: integer fields expected");
        return nullptr;
    }
    if (!h->cfg.frame_width) h->cfg.frame_width = h->cfg.coded_width;
    if (!h->cfg.frame_height) h->cfg.frame_height = h->cfg.coded_height;
    int rc = leon_create(&h->cfg, &h->d);
    if (rc != LEON_OK) {
        delete h;
        return throw_leon(env, rc);
    }
    napi_value obj;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_wrap(env, obj, h, finalize, nullptr, nullptr));
    const struct { const char* name; napi_callback fn; } methods[] = {
        {"setQuantMatrices", SetQuantMatrices}, {"acquireSlot", AcquireSlot}, {"releaseSlot", ReleaseSlot},
        {"freeDecodedSlots", FreeDecodedSlots}, {"submitPicture", SubmitPicture}, {"submitSparse", SubmitSparse}, {"convertRGBA", ConvertRGBA},
        {"readPlanes", ReadPlanes}, {"sync", Sync}, {"destroy", Destroy}};
    for (auto& m : methods) {
        napi_value fn;
        NAPI_OK(napi_create_function(env, m.name, NAPI_AUTO_LENGTH, m.fn, nullptr, &fn));
        NAPI_OK(napi_set_named_property(env, obj, m.name, fn));
    }
    return obj;
}

// ---- the native pipeline (include/leon_pipeline.h) -----------------------------------------------------
//   const p = leon.createPipeline(streamBuffer, {deviceId, parserThreads, gopsPerWindow, windowsInFlight,
//                                                maxGopPictures, loop, shardIndex, shardCount}, (window, frames, status) => {...});
//   shardIndex / shardCount: this pipeline decodes the key-map GOPs g = shardIndex (mod shardCount) on deviceId --
//   one Node process per GPU is the JavaScript host's form of the frame-parallel partition.
//   frames: [{gop, displayIndex, type, ts}] in display order; window < 0 = 'ended' (decoders/jsv.js:437).
//   p.readFrame(window, i) -> Uint8Array (copies one frame to the host: tests, thumbnails),
//   p.releaseWindow(window), p.stats(), p.info(), p.destroy().
// The callback arrives on the JavaScript thread through a napi_threadsafe_function: the pipeline's notify
// thread waits on the HIP event, nothing ever blocks the event loop (SURVEY.md 8b "Threading").
struct PipeMsg {
    int64_t window;
    int32_t status;
    std::vector<leon_pipeline_frame> frames;
};

struct PipeHandle {
    leon_pipeline* p = nullptr;
    napi_threadsafe_function tsfn = nullptr;
    napi_ref stream_ref = nullptr;          // keeps the stream's Buffer alive: the pipeline reads it in place
    std::mutex mu;
    std::map<int64_t, std::vector<leon_pipeline_frame>> out;   // delivered, not yet released
    leon_pipeline_info info{};
};

void pipe_native_cb(void* user, int64_t window, const leon_pipeline_frame* frames, int32_t n, int32_t status)
{
    PipeHandle* h = (PipeHandle*)user;
    PipeMsg* m = new PipeMsg();
    m->window = window;
    m->status = status;
    if (frames && n > 0) m->frames.assign(frames, frames + n);
    if (window >= 0) {
        std::lock_guard<std::mutex> lk(h->mu);
        h->out[window] = m->frames;
    }
    napi_call_threadsafe_function(h->tsfn, m, napi_tsfn_blocking);
}

void pipe_call_js(napi_env env, napi_value js_cb, void* ctx, void* data)
{
    PipeHandle* h = (PipeHandle*)ctx;
    PipeMsg* m = (PipeMsg*)data;
    if (env && js_cb) {
        napi_value argv[3], undef, ret;
        napi_create_int64(env, m->window, &argv[0]);
        napi_create_array_with_length(env, m->frames.size(), &argv[1]);
        for (size_t i = 0; i < m->frames.size(); i++) {
            napi_value o, v;
            napi_create_object(env, &o);
            napi_create_double(env, (double)m->frames[i].gop, &v); napi_set_named_property(env, o, "gop", v);
            napi_create_int32(env, m->frames[i].display_index, &v); napi_set_named_property(env, o, "displayIndex", v);
            napi_create_int32(env, m->frames[i].type, &v); napi_set_named_property(env, o, "type", v);
            napi_create_double(env, m->frames[i].ts_ms, &v); napi_set_named_property(env, o, "ts", v);
            napi_set_element(env, argv[1], (uint32_t)i, o);
        }
        napi_create_int32(env, m->status, &argv[2]);
        napi_get_undefined(env, &undef);
        napi_call_function(env, undef, js_cb, 3, argv, &ret);
    }
    const bool ended = m->window < 0;
    delete m;
    if (ended && h->tsfn) {          // nothing more will come: let the event loop end
        napi_release_threadsafe_function(h->tsfn, napi_tsfn_release);
        h->tsfn = nullptr;
    }
}

void pipe_finalize(napi_env env, void* data, void*)
{
    PipeHandle* h = (PipeHandle*)data;
    if (h->p) leon_pipeline_destroy(h->p);
    if (h->stream_ref) napi_delete_reference(env, h->stream_ref);
    delete h;
}

PipeHandle* pipe_unwrap(napi_env env, napi_callback_info info, size_t* argc, napi_value* argv)
{
    napi_value self;
    if (napi_get_cb_info(env, info, argc, argv, &self, nullptr) != napi_ok) return nullptr;
    PipeHandle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) != napi_ok || !h || !h->p) {
        napi_throw_error(env, nullptr, "pipeline is destroyed");
        return nullptr;
    }
    return h;
}

napi_value PipeRelease(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    PipeHandle* h = pipe_unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    int64_t w = -1;
    if (argc < 1 || napi_get_value_int64(env, argv[0], &w) != napi_ok) {
        napi_throw_type_error(env, nullptr, "releaseWindow(window)");
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->out.erase(w);
    }
    int rc = leon_pipeline_release_window(h->p, w);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value PipeReadFrame(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    PipeHandle* h = pipe_unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    int64_t w = -1;
    int32_t i = -1;
    if (argc < 2 || napi_get_value_int64(env, argv[0], &w) != napi_ok || napi_get_value_int32(env, argv[1], &i) != napi_ok) {
        napi_throw_type_error(env, nullptr, "readFrame(window, index)");
        return nullptr;
    }
    leon_pipeline_frame f{};
    {
        std::lock_guard<std::mutex> lk(h->mu);
        auto it = h->out.find(w);
        if (it == h->out.end() || i < 0 || (size_t)i >= it->second.size()) {
            napi_throw_range_error(env, nullptr, "readFrame: no such frame (window released?)");
            return nullptr;
        }
        f = it->second[(size_t)i];
    }
    const size_t bytes = (size_t)h->info.frame_width * h->info.frame_height * 4;
    napi_value ab, ta;
    void* data = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, bytes, &data, &ab));
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
    int rc = leon_pipeline_read_frame(h->p, &f, (uint8_t*)data);
    return rc == LEON_OK ? ta : throw_leon(env, rc);
}

napi_value PipeStats(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    PipeHandle* h = pipe_unwrap(env, info, &argc, nullptr);
    if (!h) return nullptr;
    leon_pipeline_stats s{};
    leon_pipeline_get_stats(h->p, &s);
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    const struct { const char* k; double val; } kv[] = {
        {"pictures", (double)s.pictures}, {"gops", (double)s.gops}, {"windows", (double)s.windows}, {"streamBytes", (double)s.stream_bytes},
        {"seconds", s.seconds}, {"parseSecondsSum", s.parse_seconds_sum}, {"uploadBytes", s.upload_bytes}, {"entries", (double)s.entries},
        {"frameWidth", (double)h->info.frame_width}, {"frameHeight", (double)h->info.frame_height},
        {"codedWidth", (double)h->info.coded_width}, {"codedHeight", (double)h->info.coded_height},
        {"pictureRate", h->info.picture_rate}, {"keyMapGops", (double)h->info.gops}, {"shardGops", (double)h->info.shard_gops}, {"firstGop", (double)h->info.first_gop}, {"duration", h->info.duration}, {"parserThreads", (double)h->info.parser_threads},
        {"gopsPerWindow", (double)h->info.gops_per_window}};
    for (auto& e : kv) {
        NAPI_OK(napi_create_double(env, e.val, &v));
        NAPI_OK(napi_set_named_property(env, o, e.k, v));
    }
    return o;
}

napi_value PipeDestroy(napi_env env, napi_callback_info info)
{
    napi_value self;
    size_t argc = 0;
    NAPI_OK(napi_get_cb_info(env, info, &argc, nullptr, &self, nullptr));
    PipeHandle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) == napi_ok && h && h->p) {
        leon_pipeline_destroy(h->p);       // joins the notify thread: no further callbacks are queued
        h->p = nullptr;
        if (h->tsfn) {
            napi_release_threadsafe_function(h->tsfn, napi_tsfn_abort);
            h->tsfn = nullptr;
        }
    }
    return nullptr;
}

// feed(validBytes): more of the stream has arrived in the Buffer the pipeline was created on
napi_value PipeFeed(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1], self;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));
    PipeHandle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) != napi_ok || !h || !h->p) { napi_throw_error(env, nullptr, "pipeline destroyed"); return nullptr; }
    int64_t v = -1;
    if (argc < 1 || napi_get_value_int64(env, argv[0], &v) != napi_ok || v < 0) { napi_throw_type_error(env, nullptr, "feed(validBytes)"); return nullptr; }
    const int rc = leon_pipeline_feed(h->p, (size_t)v);
    return rc == LEON_OK ? nullptr : throw_leon(env, rc);
}

napi_value CreatePipeline(napi_env env, napi_callback_info info)
{
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    bool is_buf = false;
    if (argc >= 1) napi_is_buffer(env, argv[0], &is_buf);
    napi_valuetype cbt = napi_undefined;
    if (argc >= 3) napi_typeof(env, argv[2], &cbt);
    if (argc < 3 || !is_buf || cbt != napi_function) {
        napi_throw_type_error(env, nullptr, "createPipeline(streamBuffer, options, callback)");
        return nullptr;
    }
    void* data = nullptr;
    size_t len = 0;
    NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &len));
    leon_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    bool ok = get_i32(env, argv[1], "deviceId", &cfg.device_id, 0) && get_i32(env, argv[1], "parserThreads", &cfg.parser_threads, 0) &&
              get_i32(env, argv[1], "gopsPerWindow", &cfg.gops_per_window, 0) && get_i32(env, argv[1], "windowsInFlight", &cfg.windows_in_flight, 0) &&
              get_i32(env, argv[1], "maxGopPictures", &cfg.max_gop_pictures, 0) && get_i32(env, argv[1], "loop", &cfg.loop, 0) &&
              get_i32(env, argv[1], "shardIndex", &cfg.shard_index, 0) && get_i32(env, argv[1], "shardCount", &cfg.shard_count, 0) &&
              get_i32(env, argv[1], "gpuParser", &cfg.gpu_parser, 0) && get_i32(env, argv[1], "displayFlavour", &cfg.display_flavour, 0);
    {   // startSeconds: begin at the key-map entry at or before this time
        napi_value v;
        bool has = false;
        if (ok && napi_has_named_property(env, argv[1], "startSeconds", &has) == napi_ok && has &&
            napi_get_named_property(env, argv[1], "startSeconds", &v) == napi_ok) {
            napi_valuetype t;
            napi_typeof(env, v, &t);
            if (t == napi_number) napi_get_value_double(env, v, &cfg.start_seconds);
        }
    }
    if (!ok) {
        napi_throw_type_error(env, nullptr, "createPipeline: integer options expected");
        return nullptr;
    }
    PipeHandle* h = new PipeHandle();
    napi_value name;
    NAPI_OK(napi_create_string_utf8(env, "leon pipeline frames", NAPI_AUTO_LENGTH, &name));
    if (napi_create_threadsafe_function(env, argv[2], nullptr, name, 0, 1, nullptr, nullptr, h, pipe_call_js, &h->tsfn) != napi_ok) {
        delete h;
        napi_throw_error(env, nullptr, "napi_create_threadsafe_function failed");
        return nullptr;
    }
    napi_create_reference(env, argv[0], 1, &h->stream_ref);
    // validBytes: the stream is still arriving (leon_pipeline_create_partial) -- the loader writes on into the same
    // Buffer and calls feed(validBytesNow), as addBuffer does in the reference (features/bitreader.js:332-430)
    // (a presence test and a double, not an int32 with -1 for "absent": 2 GiB and more of a stream may have arrived)
    bool partial = false;
    double valid = 0;
    {
        napi_value v;
        bool has = false;
        if (napi_has_named_property(env, argv[1], "validBytes", &has) == napi_ok && has && napi_get_named_property(env, argv[1], "validBytes", &v) == napi_ok) {
            napi_valuetype t;
            if (napi_typeof(env, v, &t) == napi_ok && t != napi_undefined && t != napi_null) {
                if (t != napi_number || napi_get_value_double(env, v, &valid) != napi_ok || !(valid >= 0) || valid > (double)len || valid != (double)(size_t)valid) {
                    napi_release_threadsafe_function(h->tsfn, napi_tsfn_abort);
                    napi_delete_reference(env, h->stream_ref);
                    delete h;
                    napi_throw_range_error(env, nullptr, "createPipeline: validBytes must be an integer between 0 and the stream's length");
                    return nullptr;
                }
                partial = true;
            }
        }
    }
    int rc = partial ? leon_pipeline_create_partial(&cfg, (const uint8_t*)data, len, (size_t)valid, pipe_native_cb, h, &h->p)
                     : leon_pipeline_create(&cfg, (const uint8_t*)data, len, pipe_native_cb, h, &h->p);
    if (rc != LEON_OK) {
        napi_release_threadsafe_function(h->tsfn, napi_tsfn_abort);
        napi_delete_reference(env, h->stream_ref);
        delete h;
        return throw_leon(env, rc);
    }
    leon_pipeline_get_info(h->p, &h->info);
    napi_value obj;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_wrap(env, obj, h, pipe_finalize, nullptr, nullptr));
    const struct { const char* name; napi_callback fn; } methods[] = {
        {"releaseWindow", PipeRelease}, {"readFrame", PipeReadFrame}, {"stats", PipeStats}, {"destroy", PipeDestroy}, {"feed", PipeFeed}};
    for (auto& m : methods) {
        napi_value fn;
        NAPI_OK(napi_create_function(env, m.name, NAPI_AUTO_LENGTH, m.fn, nullptr, &fn));
        NAPI_OK(napi_set_named_property(env, obj, m.name, fn));
    }
    return obj;
}

napi_value AbiVersion(napi_env env, napi_callback_info)
{
    napi_value v;
    NAPI_OK(napi_create_int32(env, leon_abi_version(), &v));
    return v;
}

napi_value Init(napi_env env, napi_value exports)
{
    if (leon_abi_version() != LEON_ABI_VERSION) {      // the addon was compiled against another include/leon.h than the library it found
        napi_throw_error(env, nullptr, "leon_napi: libleon_hip.so speaks another ABI version than this addon was built for; rebuild both");
        return exports;
    }
    napi_value fn;
    NAPI_OK(napi_create_function(env, "create", NAPI_AUTO_LENGTH, Create, nullptr, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "create", fn));
    NAPI_OK(napi_create_function(env, "createPipeline", NAPI_AUTO_LENGTH, CreatePipeline, nullptr, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "createPipeline", fn));
    NAPI_OK(napi_create_function(env, "abiVersion", NAPI_AUTO_LENGTH, AbiVersion, nullptr, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "abiVersion", fn));
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
