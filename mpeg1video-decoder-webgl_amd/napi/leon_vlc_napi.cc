// leon_vlc_napi.cc -- N-API addon over libleon_vlc.so (include/leon_vlc.h): the native bitstream
// front end for the JavaScript host.  CPU only; nothing here touches the GPU.
//
//   const vlc = require('./leon_vlc_napi.node');
//   const s = vlc.open(uint8Array, threads);      // throws on a malformed header
//   s.info() -> {frameWidth, frameHeight, codedWidth, codedHeight, mbWidth, mbHeight, groupsY, groupsC,
//                nGroups, pictureRate, duration, keymapCount, threads, intraQm, nonIntraQm}
//   s.nextPicture() -> null at the end, or {type, temporalReference, ts, newSequence, nSlices, nEntries,
//                grpOff: Uint32Array, entries: Uint32Array, qscale, intra, repadd, mvFwd, mvBwd, mbDir}
//                (typed arrays are copies: they stay valid after the next call)
//   s.seek(seconds) -> byte offset;  s.close()
#include <node_api.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/leon_vlc.h"

namespace {

struct Handle { leon_vlc_stream* s; };

napi_value fail(napi_env env, const char* what)
{
    char msg[400];
    snprintf(msg, sizeof msg, "%s: %s", what, leon_vlc_last_error());
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

Handle* unwrap(napi_env env, napi_callback_info info, size_t* argc, napi_value* argv)
{
    napi_value self;
    if (napi_get_cb_info(env, info, argc, argv, &self, nullptr) != napi_ok) return nullptr;
    Handle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) != napi_ok || !h || !h->s) {
        napi_throw_error(env, nullptr, "leon_vlc: stream is closed or invalid");
        return nullptr;
    }
    return h;
}

void finalize(napi_env, void* data, void*)
{
    Handle* h = (Handle*)data;
    if (h->s) leon_vlc_close(h->s);
    delete h;
}

napi_value copy_array(napi_env env, napi_typedarray_type type, const void* src, size_t count, size_t elem)
{
    napi_value ab, ta;
    void* data = nullptr;
    if (napi_create_arraybuffer(env, count * elem, &data, &ab) != napi_ok) return nullptr;
    if (count) memcpy(data, src, count * elem);
    if (napi_create_typedarray(env, type, count, ab, 0, &ta) != napi_ok) return nullptr;
    return ta;
}

void set_num(napi_env env, napi_value o, const char* k, double v)
{
    napi_value n;
    napi_create_double(env, v, &n);
    napi_set_named_property(env, o, k, n);
}

napi_value Info(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    Handle* h = unwrap(env, info, &argc, nullptr);
    if (!h) return nullptr;
    leon_vlc_info I;
    if (leon_vlc_get_info(h->s, &I) != LEON_VLC_OK) return fail(env, "info");
    napi_value o;
    napi_create_object(env, &o);
    set_num(env, o, "frameWidth", I.frame_width); set_num(env, o, "frameHeight", I.frame_height);
    set_num(env, o, "codedWidth", I.coded_width); set_num(env, o, "codedHeight", I.coded_height);
    set_num(env, o, "mbWidth", I.mb_width); set_num(env, o, "mbHeight", I.mb_height);
    set_num(env, o, "groupsY", I.groups_y); set_num(env, o, "groupsC", I.groups_c); set_num(env, o, "nGroups", I.n_groups);
    set_num(env, o, "hasAlpha", I.has_alpha); set_num(env, o, "pictureRate", I.picture_rate);
    set_num(env, o, "duration", I.duration); set_num(env, o, "keymapCount", I.keymap_count); set_num(env, o, "threads", I.threads);
    napi_set_named_property(env, o, "intraQm", copy_array(env, napi_uint8_array, I.intra_qm, 64, 1));
    napi_set_named_property(env, o, "nonIntraQm", copy_array(env, napi_uint8_array, I.non_intra_qm, 64, 1));
    return o;
}

napi_value NextPicture(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    Handle* h = unwrap(env, info, &argc, nullptr);
    if (!h) return nullptr;
    leon_vlc_picture p;
    const int rc = leon_vlc_next_picture(h->s, &p);
    if (rc < 0) return fail(env, "nextPicture");
    napi_value o;
    if (rc == LEON_VLC_END) {
        napi_get_null(env, &o);
        return o;
    }
    leon_vlc_info I;
    leon_vlc_get_info(h->s, &I);
    const size_t mbs = (size_t)I.mb_width * I.mb_height;
    napi_create_object(env, &o);
    set_num(env, o, "type", p.type); set_num(env, o, "temporalReference", p.temporal_reference);
    set_num(env, o, "ts", p.ts_ms); set_num(env, o, "newSequence", p.new_sequence); set_num(env, o, "openGop", p.open_gop);
    set_num(env, o, "nSlices", p.n_slices); set_num(env, o, "nEntries", p.n_entries);
    napi_set_named_property(env, o, "grpOff", copy_array(env, napi_uint32_array, p.grp_off, (size_t)p.n_groups + 1, 4));
    napi_set_named_property(env, o, "entries", copy_array(env, napi_uint32_array, p.entries, p.n_entries, 4));
    napi_set_named_property(env, o, "qscale", copy_array(env, napi_uint8_array, p.qscale, mbs, 1));
    napi_set_named_property(env, o, "intra", copy_array(env, napi_uint8_array, p.intra, mbs, 1));
    napi_value nul;
    napi_get_null(env, &nul);
    napi_set_named_property(env, o, "repadd", p.repadd ? copy_array(env, napi_uint8_array, p.repadd, mbs, 1) : nul);
    napi_set_named_property(env, o, "mvFwd", p.mv_fwd ? copy_array(env, napi_int16_array, p.mv_fwd, mbs * 2, 2) : nul);
    napi_set_named_property(env, o, "mvBwd", p.mv_bwd ? copy_array(env, napi_int16_array, p.mv_bwd, mbs * 2, 2) : nul);
    napi_set_named_property(env, o, "mbDir", p.mb_dir ? copy_array(env, napi_uint8_array, p.mb_dir, mbs, 1) : nul);
    return o;
}

napi_value Seek(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    Handle* h = unwrap(env, info, &argc, argv);
    if (!h) return nullptr;
    double t = 0;
    if (argc < 1 || napi_get_value_double(env, argv[0], &t) != napi_ok) {
        napi_throw_type_error(env, nullptr, "seek(seconds)");
        return nullptr;
    }
    uint64_t off = 0;
    if (leon_vlc_seek(h->s, t, &off) != LEON_VLC_OK) return fail(env, "seek");
    napi_value v;
    napi_create_double(env, (double)off, &v);
    return v;
}

napi_value Close(napi_env env, napi_callback_info info)
{
    size_t argc = 0;
    napi_value self;
    napi_get_cb_info(env, info, &argc, nullptr, &self, nullptr);
    Handle* h = nullptr;
    if (napi_unwrap(env, self, (void**)&h) == napi_ok && h && h->s) {
        leon_vlc_close(h->s);
        h->s = nullptr;
    }
    return nullptr;
}

napi_value Open(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 1) {
        napi_throw_type_error(env, nullptr, "open(Uint8Array[, threads])");
        return nullptr;
    }
    napi_typedarray_type type;
    size_t len;
    void* data;
    if (napi_get_typedarray_info(env, argv[0], &type, &len, &data, nullptr, nullptr) != napi_ok || type != napi_uint8_array) {
        napi_throw_type_error(env, nullptr, "open: Uint8Array expected");
        return nullptr;
    }
    int32_t threads = 0;
    if (argc > 1) napi_get_value_int32(env, argv[1], &threads);
    Handle* h = new Handle{nullptr};
    if (leon_vlc_open((const uint8_t*)data, len, threads, &h->s) != LEON_VLC_OK) {
        delete h;
        return fail(env, "open");
    }
    napi_value obj;
    napi_create_object(env, &obj);
    napi_wrap(env, obj, h, finalize, nullptr, nullptr);
    const struct { const char* name; napi_callback fn; } methods[] = {
        {"info", Info}, {"nextPicture", NextPicture}, {"seek", Seek}, {"close", Close}};
    for (auto& m : methods) {
        napi_value fn;
        napi_create_function(env, m.name, NAPI_AUTO_LENGTH, m.fn, nullptr, &fn);
        napi_set_named_property(env, obj, m.name, fn);
    }
    return obj;
}

napi_value Init(napi_env env, napi_value exports)
{
    if (leon_vlc_abi_version() != LEON_VLC_ABI_VERSION) {      // built against another include/leon_vlc.h than the library it found
        napi_throw_error(env, nullptr, "leon_vlc_napi: libleon_vlc.so speaks another ABI version than this addon was built for; rebuild both");
        return exports;
    }
    napi_value fn;
    napi_create_function(env, "open", NAPI_AUTO_LENGTH, Open, nullptr, &fn);
    napi_set_named_property(env, exports, "open", fn);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
