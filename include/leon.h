/*
 * leon.h -- C ABI of the MI355X-native macroblock-reconstruction path
 *           (libleon_hip.so; gfx950 only).
 *
 * This is the drop-in boundary for the ONE hot path of the reference decoder:
 * everything the reference does between "the slice loop has filled the
 * per-picture arrays" and "decoded planes are on the GPU / an RGBA frame is on
 * the canvas".  Each entry point names the reference interface it replaces
 * (paths under /root/reference).  Plain pointers and sizes only; no torch, no
 * HIP types.  There is NO CPU fallback: every call fails with LEON_ERR_NO_DEVICE
 * when no gfx950 device is usable.
 *
 * Data handed over per picture -- exactly the arrays
 * jsv.prototype.IDCT_GL uploads with texImage2D (decoders/jsv.js:1204-1298):
 *   coef_y/cb/cr  int16 LE dense planes of RAW quantised levels (intra DC in the
 *                 0..255 predictor domain), natural order, at the pixel position
 *                 of their block; row stride = plane width   (jsv.js:1237-1243)
 *   qscale        u8 [mbH][mbW] quantiser_scale per macroblock (jsv.js:1204-1206)
 *   intra         u8 [mbH][mbW] non-zero = intra macroblock    (jsv.js:1215-1217)
 *   repadd        u8 [mbH][mbW] >=128 = replace (no prediction) (jsv.js:1282-1284)
 *   mv_fwd        int16 [mbH][mbW][2] (H,V) luma half-pel units (jsv.js:1296-1298)
 *   mv_bwd,mb_dir B pictures only (beyond the reference, ISO 11172-2 2.4.4.3):
 *                 backward vectors, and per-MB direction bits 1=fwd 2=bwd
 * Arithmetic domain: |level| <= 32767, qscale 0..31, matrix entries 0..255.
 *
 * Threading: the calls on ONE decoder must be serialised by the caller (the reference drives its
 * GL context from one thread, decoders/jsv.js:427-430); different decoders are independent and may
 * be used from different threads.  Everything a decoder launches is ordered on its HIP stream;
 * submits return as soon as the work is queued.
 */
#ifndef LEON_H
#define LEON_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 4): leon_config.reserved became contiguous_slots, leon_pipeline_config.gpu_parser 0 now means the GPU parser
 * (leon_pipeline.h), leon_device_pool_stats, leon_measure_stream_bandwidth; leon_vlc.h carries a version of its own. */
#define LEON_ABI_VERSION 3

enum {
    LEON_OK = 0,
    LEON_ERR_INVALID = -1,      /* bad argument (shape, slot, type) */
    LEON_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP failure at create */
    LEON_ERR_HIP = -3,          /* a HIP call failed; see leon_last_error() */
    LEON_ERR_NO_FREE_SLOT = -4, /* = throw "no free render buffers" (jsv.js:1175) */
    LEON_ERR_NOMEM = -5
};

/* pictureCodingType values of the reference (decoders/jsv.js PICTURE_TYPE_*) */
enum { LEON_PIC_I = 1, LEON_PIC_P = 2, LEON_PIC_B = 3 };

/* where the pointers of a leon_picture / rgba destination live */
enum { LEON_MEM_HOST = 0, LEON_MEM_DEVICE = 1 };

/* colour conversion flavour of leon_convert_rgba */
enum {
    LEON_RGB_CPU_TWIN = 0, /* fp64, = jsv.prototype.YCbCrToRGBA player/easybits.player.js:2674-2785 (parity target) */
    LEON_RGB_GL = 1        /* fp32 matrix, = SHADER_FRAGMENT_YCBCRTORGBA player/parts/end.js:77-156 */
};

typedef struct leon_decoder leon_decoder;
typedef struct leon_batch leon_batch;

typedef struct leon_config {
    int32_t coded_width;   /* mbWidth<<4  (jsv.js:364) */
    int32_t coded_height;  /* mbHeight<<4 (jsv.js:365) */
    int32_t frame_width;   /* display crop (jsv.js:360-361, player.js:820-821) */
    int32_t frame_height;
    int32_t n_slots;       /* output ring; the reference uses 13 (jsv.js:24, :58-73) */
    int32_t device_id;     /* HIP device ordinal */
    void*   stream;        /* hipStream_t to run on, or NULL: the decoder owns one */
    int32_t alpha;         /* ABI 2: yuva stream (container flag `a`, decoders/jsv.js:256-259): every slot gets a fourth,
                              luma-sized plane [Y|Cb|Cr|A] like the reference's 4-plane ring (jsv.js:59-73); pictures
                              carry coef_a; RGBA output takes its A byte from the plane.  The reference allocates the
                              plane and never decodes it (IDCT_GL loops over three components, jsv.js:1223): A is
                              reconstructed like luma here -- same maps, luma vectors, same matrices.  Needs an even
                              frame width. */
    int32_t contiguous_slots; /* 1: ask for a physically contiguous slot ring (leon_device_malloc explains what that buys: the
                              launches that read references run at their fast end, bench.py asks for it).  0, the default:
                              an ordinary allocation.  A contiguous ring is taken from, and at leon_destroy returned to, a
                              pool that lives as long as the process: contiguous memory is never handed back to the driver
                              (round 3 saw wrong B pictures in LATER pipelines of processes that had hipFree'd contiguous
                              rings -- 13 of 30 whole-suite runs, none when the rings were never freed; leon_hip.cpp
                              big_alloc).  So the knob is safe in a process that creates and destroys many decoders; what
                              it costs is that the process keeps the high-water mark of its contiguous requests
                              (leon_device_pool_stats).  (Was `reserved` in ABI 2.) */
} leon_config;

typedef struct leon_picture {
    int32_t type;          /* LEON_PIC_I / _P / _B */
    int32_t out_slot;      /* slot to write (from leon_acquire_slot or caller-managed) */
    int32_t ref_fwd_slot;  /* P, B: forward reference = prev_pic_framebuffer (jsv.js:665, :1320) */
    int32_t ref_bwd_slot;  /* B only */
    const int16_t* coef_y;
    const int16_t* coef_cb;
    const int16_t* coef_cr;
    const uint8_t* qscale;
    const uint8_t* intra;
    const uint8_t* repadd; /* P, B */
    const int16_t* mv_fwd; /* P, B */
    const int16_t* mv_bwd; /* B */
    const uint8_t* mb_dir; /* B */
    /* Fused display conversion (ABI 2; optional, NULL = off).  rgba_out is a DEVICE pointer to
     * frame_width*frame_height*4 bytes, 16-byte aligned: the reconstruction kernel itself converts the
     * picture (= renderFrameGL / YCbCrToRGBA, player/easybits.player.js:2787-2858 / :2674-2785, the
     * LEON_RGB_CPU_TWIN arithmetic) instead of a later leon_convert_rgba reading the planes back.
     * Needs frame_width % 8 == 0.  (yuva decoders: the A byte of every pixel comes from the A component.)  no_planes != 0: the slot's planes are not written at all -- for a
     * picture nobody predicts from (B pictures); out_slot is then ignored (-1 is accepted). */
    void*   rgba_out;
    int32_t no_planes;
    int32_t qm_set;        /* ABI 3 (was `reserved`, 0): which of the decoder's quantiser-matrix sets the picture is dequantised
                              with -- 0 = the matrices of leon_set_quant_matrices, > 0 = a set from leon_add_quant_matrices.
                              The reference reloads both matrices at every sequence header (decoders/jsv.js:540-558); a
                              batch of pictures from different sequences (the pipeline's windows of GOP shards) names one
                              set per picture. */
    const int16_t* coef_a; /* yuva decoders (leon_config.alpha): the A plane's raw levels, coded-luma size; else NULL */
} leon_picture;

/* The same picture with its coefficients as sparse per-group lists -- the format the native
 * bitstream front end emits (include/leon_vlc.h) -- instead of the three dense
 * planes the reference uploads.  Equivalent to a leon_picture whose planes hold `level` at every
 * listed position and 0 elsewhere.  Lists and maps live where `mem` of the call says. */
typedef struct leon_sparse_picture {
    int32_t type, out_slot, ref_fwd_slot, ref_bwd_slot;
    const uint32_t* grp_off;   /* [n_groups+1], n_groups = 2*mbH*ceil(2*mbW/8) + 2*mbH*ceil(mbW/8); yuva decoders: the A
                                  plane's 2*mbH*ceil(2*mbW/8) groups follow, numbered like the luma groups */
    const uint32_t* entries;   /* [n_entries]: (tile byte offset r*128+b*16+c*2) << 16 | (uint16)level */
    uint32_t n_entries;
    int32_t reserved;
    const uint8_t* qscale;
    const uint8_t* intra;
    const uint8_t* repadd;
    const int16_t* mv_fwd;
    const int16_t* mv_bwd;
    const uint8_t* mb_dir;
    void*   rgba_out;          /* fused display conversion, as in leon_picture */
    int32_t no_planes;
    int32_t qm_set;            /* as in leon_picture (was `reserved2`) */
} leon_sparse_picture;

typedef struct leon_kernel_stats {
    uint64_t launches;        /* timed launches since leon_timing_reset */
    double   total_ms;        /* sum of their HIP-event durations */
    double   algorithmic_bytes; /* sum over launches of SURVEY.md 8d bytes, priced per macroblock from the picture's
                                   own maps: I 1154; P 1158 + 384 if predicted (= 8d's 1542); B 1162 + 384 per
                                   direction used (1546 one-sided, 1930 bidirectional = 8d's figure).  Sparse
                                   launches: 4 B per entry and per group offset in place of the 768 B/MB of
                                   dense coefficients */
    uint64_t macroblocks;     /* sum over launches */
} leon_kernel_stats;

/* one timed launch (leon_timing_get_launches) */
typedef struct leon_launch_time {
    int32_t kind;             /* 0 = reconstruction kernel, 1 = colour conversion */
    int32_t pic_type;         /* reconstruction launches: LEON_PIC_I / _P / _B; else 0 */
    double  ms;               /* HIP-event duration */
    double  algorithmic_bytes;
    uint64_t macroblocks;
} leon_launch_time;

int leon_abi_version(void);

/* thread-local text of the last error returned on this thread */
const char* leon_last_error(void);

/* = new jsv_dec + decoder._initGL(gl) + initGLBuffers (player/easybits.player.js:584-585,
 *   decoders/jsv.js:88-208, :51-87): allocates the slot ring and uploads the default
 *   quant matrices + premultiplier (jsv.js:139-150). */
int leon_create(const leon_config* cfg, leon_decoder** out);
void leon_destroy(leon_decoder* d);

/* = the QUANT_MATRIX re-uploads on a sequence header (decoders/jsv.js:540-558).
 *   64 bytes each, natural (de-zig-zagged) order; NULL keeps the current one. */
int leon_set_quant_matrices(leon_decoder* d, const uint8_t* intra64, const uint8_t* non_intra64);
/* ABI 3.  The same for a caller that decodes pictures of SEVERAL sequences in one batch (the pipeline: every key-map GOP
 *   starts with a sequence header of its own, and the reference reloads the matrices at each, jsv.js:540-558): registers
 *   a further set of matrices beside set 0 (what leon_set_quant_matrices writes) and returns its id in *set, the same id
 *   for the same matrices; pictures name their set in leon_picture.qm_set.  NULL = the default matrix (jsv.js:1777-1806).
 *   A set never changes once registered; at most 255 per decoder.  Asynchronous: ordered by the decoder's stream. */
int leon_add_quant_matrices(leon_decoder* d, const uint8_t* intra64, const uint8_t* non_intra64, int32_t* set);

/* = jsv.prototype.setRenderBuffer (decoders/jsv.js:1165-1176): first free slot, marked in use */
int leon_acquire_slot(leon_decoder* d, int32_t* slot);
/* = texture.inuse = false in renderFrameGL (player/easybits.player.js:2820) */
int leon_release_slot(leon_decoder* d, int32_t slot);
/* = jsv.prototype.GLfreeDecodedBuffers (decoders/jsv.js:1160-1164), used by seek (:1623) */
int leon_free_decoded_slots(leon_decoder* d);

/* = jsv.prototype.IDCT_GL (decoders/jsv.js:1177-1336) for one picture whose arrays
 *   are in host memory: staged to the device and reconstructed asynchronously on the
 *   decoder's stream.  The arrays may be reused as soon as the call returns. */
int leon_submit_picture(leon_decoder* d, const leon_picture* pic);

/* Batched IDCT_GL: n mutually independent pictures (no picture of the batch writes a slot that
 * another one writes or reads; checked, LEON_ERR_INVALID otherwise).
 * mem = LEON_MEM_DEVICE: every pointer in pics[] is a device pointer (resident boundary tensors;
 * the arrays must stay valid until the work completed); one kernel launch per picture type present.
 * mem = LEON_MEM_HOST: each picture is staged and launched like leon_submit_picture (n launches);
 * the batched host path is leon_pipeline_* below. */
int leon_submit_batch(leon_decoder* d, const leon_picture* pics, int32_t n, int32_t mem);

/* A prepared batch keeps its descriptors on the device so that re-running it costs
 * one kernel launch and no copies (device-resident pictures only). */
int leon_batch_create(leon_decoder* d, const leon_picture* pics, int32_t n, leon_batch** out);
int leon_batch_run(leon_decoder* d, const leon_batch* b);
void leon_batch_destroy(leon_decoder* d, leon_batch* b);

/* The three calls above for the sparse boundary.  A batch is either all dense or all sparse.
 * mem = LEON_MEM_HOST: lists and maps are staged per picture (n_entries*4 + offsets instead of
 * the 6.27 MB of dense planes a 1080p picture uploads, decoders/jsv.js:1237-1243). */
int leon_submit_sparse(leon_decoder* d, const leon_sparse_picture* pics, int32_t n, int32_t mem);
int leon_batch_create_sparse(leon_decoder* d, const leon_sparse_picture* pics, int32_t n, leon_batch** out);

/* = renderFrameGL(_frame) (player/easybits.player.js:2787-2858) / YCbCrToRGBA (:2674-2785):
 *   slot -> RGBA8 frame_width x frame_height, tightly packed.  dst_mem says where
 *   rgba lives.  Does NOT release the slot (call leon_release_slot). */
int leon_convert_rgba(leon_decoder* d, int32_t slot, void* rgba, int32_t dst_mem, int32_t flavour);
/* n slots -> n consecutive RGBA frames in device memory, one launch */
int leon_convert_rgba_batch(leon_decoder* d, const int32_t* slots, int32_t n, void* rgba_device, int32_t flavour);

/* test / read-back helpers (the reference never reads planes back; dumpPixels2
 * decoders/jsv.js:1141-1159 is its only analogue) */
int leon_read_planes(leon_decoder* d, int32_t slot, uint8_t* y, uint8_t* cb, uint8_t* cr);
int leon_write_planes(leon_decoder* d, int32_t slot, const uint8_t* y, const uint8_t* cb, const uint8_t* cr);
/* the fourth plane of a yuva decoder's slot (coded-luma size) */
int leon_read_alpha_plane(leon_decoder* d, int32_t slot, uint8_t* a);
int leon_write_alpha_plane(leon_decoder* d, int32_t slot, const uint8_t* a);
/* device address of a slot's [Y|Cb|Cr] planes (for zero-copy consumers) */
int leon_slot_device_ptr(leon_decoder* d, int32_t slot, void** ptr, size_t* bytes);

/* wait for everything submitted so far (both streams, see leon_set_overlap_convert) */
int leon_sync(leon_decoder* d);

/* Overlap display conversion with reconstruction: when on, leon_convert_rgba_batch with a device
 * destination runs on a second HIP stream, ordered after the reconstruction work submitted so
 * far, and the following submits do not wait for it -- unless one of them writes a slot whose
 * conversion is still pending, in which case the library orders it behind that conversion.
 * (The reference has one GL queue; this is the MI355X-side equivalent of its decode-ahead while
 * requestAnimationFrame renders, player/easybits.player.js:2478-2504.)  Default: off. */
int leon_set_overlap_convert(leon_decoder* d, int32_t on);

/* HIP-event timing of the kernels launched by this decoder (bench.py roofline).
 * kind: 0 = reconstruction kernel (dequant+IDCT+MC), all picture types; 1 = colour conversion;
 *       2 / 3 / 4 = the reconstruction launches of I / P / B pictures only. */
int leon_timing_enable(leon_decoder* d, int32_t on);
int leon_timing_reset(leon_decoder* d);
int leon_timing_get(leon_decoder* d, int32_t kind, leon_kernel_stats* out);
/* the timed launches one by one, in submission order: up to `cap` records into out[], *n = how many there are
 * (bench.py: the spread of a launch class, and the one-sided and the mixed B launches apart) */
int leon_timing_get_launches(leon_decoder* d, leon_launch_time* out, int32_t cap, int32_t* n);

/* Device memory for a caller's large, long-lived buffers: physically CONTIGUOUS where the device has it
 * (hipExtMallocWithFlags, hipDeviceMallocContiguous), an ordinary allocation otherwise.
 * Why a caller should care: the reconstruction launches stream through many buffers at once, and the same launch on the
 * same box took 0.39 ... 0.56 ms depending on which physical pages its RGBA frames had drawn -- an ordinary allocation
 * is built from whatever fragments are free; with contiguous memory the runs of one box agree to better than 1 %
 * (round 3, tools/probe/spread_probe.py, profiles/r03_launch_spread.json; the mechanism is not established).  The boundary
 * tensors and RGBA frames a caller hands to leon_submit_* may come from anywhere; bench.py takes them from here (and
 * sets leon_config.contiguous_slots).  The library's own rings are ordinary allocations unless asked (contiguous_slots).
 * (The reference's counterpart: gl.createTexture / texImage2D storage, jsv.js:51-87.)
 * leon_device_malloc: on device `device_id` (which becomes the calling thread's current device); *contiguous (may be
 * NULL) = 1 when the contiguous request was granted.  Requests of 1 MiB and more are rounded up to 2 MiB granules.
 * leon_device_free: waits for the device like hipFree, then hands a contiguous buffer to the process's POOL -- the next
 * contiguous request of that device reuses it; it is never returned to the driver (see contiguous_slots) -- and frees
 * an ordinary one.  leon_device_free(NULL) is a no-op.
 * leon_device_pool_stats: bytes of contiguous memory this process has taken from the driver (all devices), how many of
 * them are handed out right now, and in how many driver allocations; any pointer may be NULL. */
int leon_device_malloc(int32_t device_id, size_t bytes, void** ptr, int32_t* contiguous);
int leon_device_free(void* ptr);
int leon_device_pool_stats(uint64_t* held_bytes, uint64_t* in_use_bytes, int32_t* segments);

/* measured device copy bandwidth (GB/s) over `bytes` with a streaming float4
 * copy kernel: the "measured HBM roofline" of BASELINE.md section 2 */
int leon_measure_copy_bandwidth(leon_decoder* d, size_t bytes, int32_t iters, double* gbps);
/* The same 16-byte-per-lane shape with `reads` source streams and `writes` destination streams of `bytes` each (0..2 and
 * 0..2, not both 0): (0,1) = what the memory system takes when a launch only writes, (1,2) a launch that writes twice what
 * it reads -- the fused I and P launches write 55-65 % of their bytes, and a 1:1 copy says little about how close they are
 * to their wall.  *gbps counts every stream's bytes.  Buffers as in leon_measure_copy_bandwidth (the caller's allocator). */
int leon_measure_stream_bandwidth(leon_decoder* d, size_t bytes, int32_t iters, int32_t reads, int32_t writes, double* gbps);

#ifdef __cplusplus
}
#endif
#endif
