/*
 * leon_pipeline.h -- native decode pipeline over libleon_vlc + libleon_hip (exported by libleon_hip.so).
 *
 * What it replaces in the reference: the page's decode loop -- decodeFrame() pulling one picture at
 * a time through the bit-serial parser and one IDCT_GL per picture, on the page's only thread
 * (decoders/jsv.js:426-469, :1593-1599, :1177-1336; player/easybits.player.js:2310-2324, :2543-2617).
 * Here the same stream bytes go through
 *     K parser threads   one libleon_vlc stream per GOP shard (cut at the key map, decoders/jsv.js:264-350;
 *                        closed GOPs share nothing), lists and maps written straight into pinned memory
 *     one submit thread  a window of W consecutive GOPs at a time: one asynchronous upload per GOP, then ONE
 *                        kernel launch per picture type and dependency level ACROSS the window's GOPs, the
 *                        display conversion fused in (leon_picture.rgba_out); B pictures write no planes
 *     one notify thread  waits on the window's HIP event and calls back with the frames (RGBA8 in device
 *                        memory, display order) -- no caller thread ever blocks on the GPU
 * No interpreter is in the loop; a Node host learns of frames through a napi_threadsafe_function
 * (mpeg1video-decoder-webgl_amd/napi/leon_napi.cc), the MI355X-side of the reference's 'frame' event
 * (decoders/jsv.js:673).
 *
 * Requirements on the stream: JSV with a key map (or a raw elementary stream: one shard), closed GOPs, ONE picture size
 * (a sequence header that changes the size ends the run with an error; one that changes the quantiser matrices -- the
 * reference reloads them at every header, decoders/jsv.js:540-558 -- is honoured per picture, round 4).  Any frame width:
 * widths that are no multiple of 8 take an unfused road inside (planes + one conversion launch per picture).
 */
#ifndef LEON_PIPELINE_H
#define LEON_PIPELINE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct leon_pipeline leon_pipeline;

#define LEON_PIPELINE_PARSER_DEFAULT 0
#define LEON_PIPELINE_PARSER_GPU 1
#define LEON_PIPELINE_PARSER_HOST (-1)

typedef struct leon_pipeline_config {
    int32_t device_id;
    int32_t parser_threads;     /* K; <= 0: one per hardware thread, at most 16 */
    int32_t gops_per_window;    /* W: independent GOPs decoded together = pictures per launch and type; <= 0: 32 */
    int32_t windows_in_flight;  /* RGBA / staging rings; <= 0: 2 (3 with gpu_parser) */
    int32_t max_gop_pictures;   /* frames reserved per GOP in a window; <= 0: the longest GOP of the stream (counted at create) */
    int32_t loop;               /* benchmarking: decode the stream this many times over (GOP ids keep counting); <= 0: once */
    /* Frame-parallel GOP shards across the GPUs of a node: this pipeline decodes the key-map GOPs
     * g with g % shard_count == shard_index only -- one process (or pipeline) per GPU, each with its device_id,
     * all given the same stream; nothing is exchanged between them (closed GOPs share nothing, the key map is
     * the stream's own index, decoders/jsv.js:264-350).  shard_count <= 1: everything. */
    int32_t shard_index, shard_count;
    /* = jsv.prototype.seek (decoders/jsv.js:1618-1648) at start-up: begin with the key-map entry at or before this
     * time (seconds) instead of the first one; 0 = from the start.  (A running pipeline is not repositioned: a
     * host seeks by destroying it and creating one at the new time, as the reference frees all its output
     * buffers on a seek, jsv.js:1623.) */
    double start_seconds;
    /* Where the slice layer is decoded (everything below a slice start code: macroblock headers, vectors, coefficients
     * -- decodeSlice .. decodeBlockGL, decoders/jsv.js:683-1525):
     *   LEON_PIPELINE_PARSER_DEFAULT (0) and LEON_PIPELINE_PARSER_GPU (1): on the GPU, one lane per slice
     *     (csrc/leon_vlc_gpu.h); the parser threads only read the picture layer and upload the stream bytes.  Errors of
     *     a slice surface when its window completes.  The default since round 3: six times the host front end.
     *     A stream beyond the GPU parser's limits (a picture of more than ~40 k block groups -- larger than 4096 x 2304 --,
     *     a GOP shard of 2^28 bytes) is decoded on the parser threads under DEFAULT and refused under an explicit GPU
     *     (leon_pipeline_info.gpu_parser says which it is).  A picture whose slices OVERLAP (MPEG-1 forbids it; the
     *     reference decodes them one after the other, the later one wins) is refused by the GPU parser, which decodes a
     *     picture's slices side by side, when its window completes: the host parser decodes such a stream.
     *   LEON_PIPELINE_PARSER_HOST (-1): on the parser threads (libleon_vlc.so).
     * Same frames either way. */
    int32_t gpu_parser;
    /* Arithmetic of the frames' colour conversion, leon.h's LEON_RGB_*:
     *   LEON_RGB_CPU_TWIN (0): the integer-exact twin of the reference's CPU conversion, fused into the reconstruction launches
     *     (what every other path of this library delivers; bit-exact to the oracle).
     *   LEON_RGB_GL (1): the fp32 arithmetic of the reference's LIVE display -- renderFrameGL drawing with
     *     SHADER_FRAGMENT_YCBCRTORGBA (player/easybits.player.js:2787-2858, player/parts/end.js:77-156) -- for a host that
     *     wants the pixels the page shows (within 1 LSB of the executed reference's canvas, tests/test_pipeline_gl_flavour_gpu.py).
     *     Takes the unfused road: every picture writes its planes, one k_rgba_gl launch per picture.  Not with yuva streams. */
    int32_t display_flavour;
} leon_pipeline_config;

/* One decoded picture.  rgba stays valid until leon_pipeline_release_window(window) */
typedef struct leon_pipeline_frame {
    uint64_t gop;               /* GOP id (key-map index, counting on across loops) */
    int32_t  display_index;     /* temporal reference inside its GOP */
    int32_t  type;              /* LEON_PIC_I / _P / _B */
    double   ts_ms;             /* presentation time: GOP time code + display_index / picture rate */
    void*    rgba;              /* DEVICE pointer: frame_width * frame_height * 4 bytes */
} leon_pipeline_frame;

/* Called on the pipeline's notify thread once per window, frames in display order (GOP-major).
 * n_frames == 0 with window < 0 signals the end of the stream ('ended', decoders/jsv.js:437);
 * status != 0 an error (leon_pipeline_error() has the text).
 * Lifetimes: the `frames` ARRAY belongs to the window -- it is valid until the callback returns or the window is
 * released, whichever comes first (a callback that releases the window must have read frames[] before; copy what is
 * kept); the device memory `rgba` points to stays valid until leon_pipeline_release_window(window).  A window
 * delivered with status != 0 (n_frames may be 0) holds its ring entry and staging like any other and must be
 * released too. */
typedef void (*leon_pipeline_callback)(void* user, int64_t window, const leon_pipeline_frame* frames, int32_t n_frames, int32_t status);

typedef struct leon_pipeline_info {
    int32_t coded_width, coded_height, frame_width, frame_height;
    double  picture_rate, duration;
    uint32_t gops;              /* key-map entries (1 for a stream without key map) */
    uint32_t shard_gops;        /* how many of them this pipeline decodes (per pass over the stream) */
    uint32_t first_gop;         /* key-map entry the run starts with (start_seconds) */
    int32_t parser_threads, gops_per_window;
    int32_t gpu_parser;         /* 1: the slice layer is decoded on the GPU, 0: on the parser threads (what LEON_PIPELINE_PARSER_DEFAULT chose) */
    int32_t display_flavour;    /* LEON_RGB_CPU_TWIN / LEON_RGB_GL, as configured */
} leon_pipeline_info;

typedef struct leon_pipeline_stats {
    uint64_t pictures, gops, windows, stream_bytes;
    double   seconds;           /* first parser start -> last window completed (so far) */
    double   parse_seconds_sum; /* summed over the parser threads */
    double   upload_bytes;      /* what crossed PCIe */
    uint64_t entries;           /* non-zero coefficients decoded (host front end; 0 with gpu_parser: they stay on the device) */
} leon_pipeline_stats;

/* Copies nothing: `stream` must stay valid until leon_pipeline_destroy.  Starts decoding at once. */
int leon_pipeline_create(const leon_pipeline_config* cfg, const uint8_t* stream, size_t bytes,
                         leon_pipeline_callback cb, void* user, leon_pipeline** out);
/* The same for a stream that is still arriving -- the reference's decoder consumes a growing buffer, stalls when it runs
 * dry and resumes when a chunk is appended (features/bitreader.js:332-430 addBuffer, :135-189 has; decoders/jsv.js
 * :426-469): `stream` is the buffer for the WHOLE file (`bytes` = its final size, known from the HTTP headers as in the
 * reference's {data, start, end, total} chunks), of which the first `valid_bytes` are there -- enough for the container
 * header, the key map and the first sequence header.  The loader writes on into the same buffer and reports progress
 * with leon_pipeline_feed(p, valid_bytes_now); a GOP is parsed once the bytes up to the next key-map entry have
 * arrived, windows are delivered as they complete.  With max_gop_pictures <= 0 a partial stream reserves 16 frames per GOP. */
int leon_pipeline_create_partial(const leon_pipeline_config* cfg, const uint8_t* stream, size_t bytes, size_t valid_bytes,
                                 leon_pipeline_callback cb, void* user, leon_pipeline** out);
int leon_pipeline_feed(leon_pipeline* p, size_t valid_bytes);
int leon_pipeline_get_info(leon_pipeline* p, leon_pipeline_info* out);
/* the consumer is done with a window's frames: its RGBA ring entry and staging may be reused */
int leon_pipeline_release_window(leon_pipeline* p, int64_t window);
/* blocks until every window has been delivered and the final callback (window -1) has returned; returns the
 * first error.  Not to be called from inside the callback. */
int leon_pipeline_wait(leon_pipeline* p);
int leon_pipeline_get_stats(leon_pipeline* p, leon_pipeline_stats* out);
/* copy one frame of a delivered, not yet released window to host memory (tests, thumbnails) */
int leon_pipeline_read_frame(leon_pipeline* p, const leon_pipeline_frame* f, uint8_t* rgba_host);
const char* leon_pipeline_error(leon_pipeline* p);
void leon_pipeline_destroy(leon_pipeline* p);

#ifdef __cplusplus
}
#endif
#endif
