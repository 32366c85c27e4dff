/*
 * leon_vlc.h -- C ABI of the native bitstream front end (libleon_vlc.so; plain C++17, no GPU).
 *
 * the reference parses the stream with a bit-serial tree walk on one
 * JavaScript thread (readCode decoders/jsv.js:1593-1599, getBits decoders/bitreader.js:443-540,
 * decodeBlockGL :1338-1525) and uploads 6.27 MB of dense int16 planes per 1080p picture
 * (:1237-1243).  This library replaces that layer with a table-driven multi-bit parser that
 * decodes the slices of a picture on worker threads (slices are independently decodable after
 * their start code) and emits the coefficients as SPARSE per-group lists -- what
 * leon_submit_sparse() of include/leon.h consumes -- next to the same per-macroblock maps the
 * reference uploads.  Same stream semantics as the JavaScript mirror of the reference parser
 * (mpeg1video-decoder-webgl_amd/js/jsv_decoder.js), which is pinned to the reference by
 * tests/golden/parser_*.json; B pictures are read too (ISO/IEC 11172-2).
 *
 * Sparse coefficient format ("group lists"):
 *   A group is 8 horizontally adjacent 8x8 blocks (64x8 samples) of one plane -- the unit one
 *   GPU wave reconstructs.  groups_y = ceil(coded_width/64) groups per luma block row,
 *   groups_c = ceil(coded_width/128) per chroma block row.  Group ids:
 *     luma   block row R (0 .. 2*mb_height-1), group g :  R*groups_y + g
 *     Cb     block row R (0 .. mb_height-1)            :  n_y + R*groups_c + g
 *     Cr                                               :  n_y + n_c + R*groups_c + g
 *   with n_y = 2*mb_height*groups_y, n_c = mb_height*groups_c, n_groups = n_y + 2*n_c.
 *   yuva streams (container flag `a` = 1, leon_vlc_info.has_alpha): the A component -- four more blocks per
 *   macroblock, placed and coded like the luma blocks (syntax: tools/jsv_writer.py write_picture; the reference
 *   reads the flag, decoders/jsv.js:256-259, and defines no syntax) -- adds n_y groups numbered like the luma
 *   ones:   A  block row R, group g :  n_y + 2*n_c + R*groups_y + g      (n_groups = 2*n_y + 2*n_c)
 *   grp_off[n_groups+1] are prefix offsets into entries[]; the entries of a group are
 *   contiguous, in no particular order.  One entry = one non-zero level:
 *     bits  0..15  level (int16; intra DC in the 0..255 predictor domain, jsv.js:1346-1443)
 *     bits 16..25  byte offset of the coefficient in the group's int16 tile [row r][block b][col c]
 *                  = r*128 + b*16 + c*2   (r, c natural order inside the block; b = block in group)
 *
 * Threading: one stream is driven from one thread at a time (it owns its worker threads);
 * different streams are independent -- one per GOP shard is how the front end scales across cores.
 */
#ifndef LEON_VLC_H
#define LEON_VLC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4; the header had no version before): leon_vlc_picture has open_gop, leon_vlc_picture_scan has end_byte /
 * open_gop / reserved -- a binding compiled against an older header would hand the library structs that are too small.
 * Bindings check leon_vlc_abi_version() == LEON_VLC_ABI_VERSION at load. */
#define LEON_VLC_ABI_VERSION 2

enum {
    LEON_VLC_OK = 0,
    LEON_VLC_END = 0,            /* leon_vlc_next_picture: end of stream */
    LEON_VLC_PICTURE = 1,        /* leon_vlc_next_picture: *out is filled */
    LEON_VLC_ERR_INVALID = -1,   /* bad argument */
    LEON_VLC_ERR_STREAM = -2,    /* malformed stream (invalid code, coefficient index overflow ...) */
    LEON_VLC_ERR_NOMEM = -3
};

typedef struct leon_vlc_stream leon_vlc_stream;

typedef struct leon_vlc_info {
    int32_t frame_width, frame_height;     /* sequence header (jsv.js:491-497) */
    int32_t coded_width, coded_height;     /* mb_width<<4, mb_height<<4 (jsv.js:364-365) */
    int32_t mb_width, mb_height;
    int32_t groups_y, groups_c, n_groups;
    int32_t has_alpha;                     /* 'a' flag of the long container header (jsv.js:256-259); -1: short header, no flag */
    double  picture_rate;
    double  duration;                      /* seconds, container header */
    uint32_t keymap_count;                 /* START_MAP entries (jsv.js:264-268) */
    uint32_t threads;                      /* worker threads in use (including the caller) */
    uint8_t intra_qm[64], non_intra_qm[64];/* natural order; defaults when the stream has none */
} leon_vlc_info;

typedef struct leon_vlc_picture {
    int32_t type;                 /* 1 I, 2 P, 3 B */
    int32_t temporal_reference;
    double  ts_ms;                /* GOP time code of the first picture after a GOP header, else 0 (jsv.js:471-489) */
    int32_t new_sequence;         /* 1: a sequence header preceded this picture (matrices in leon_vlc_info may differ) */
    int32_t n_groups;
    uint32_t n_entries;
    const uint32_t* grp_off;      /* [n_groups+1] */
    const uint32_t* entries;      /* [n_entries] */
    const uint8_t* qscale;        /* [mb_height*mb_width], persists across pictures (jsv.js:391-393) */
    const uint8_t* intra;
    const uint8_t* repadd;        /* P, B; else NULL */
    const int16_t* mv_fwd;        /* P, B */
    const int16_t* mv_bwd;        /* B */
    const uint8_t* mb_dir;        /* B */
    uint32_t n_slices;
    int32_t open_gop;             /* 1: the first picture behind a GOP header whose closed_gop bit is 0 -- B pictures in
                                     front of the GOP's second anchor may predict from the GOP before it */
} leon_vlc_picture;

int leon_vlc_abi_version(void);
const char* leon_vlc_last_error(void);

/* Copies `n` bytes of a JSV stream (container header + key map, decoders/jsv.js:237-313) or of a
 * raw MPEG-1 video elementary stream (starts with 00 00 01 B3), or of a GOP shard of a JSV stream
 * (starts with its sequence header 00 00 01 C3, see leon_vlc_get_keymap).  threads <= 0: one per hardware
 * thread, at most 16.  Reads up to the first sequence header so that leon_vlc_get_info is valid. */
int leon_vlc_open(const uint8_t* data, size_t n, int32_t threads, leon_vlc_stream** out);
/* The same for a GOP shard of a stream whose container header is known to the caller: has_alpha = the
 * leon_vlc_info.has_alpha of the whole stream (the `a` flag is in the container header, jsv.js:256-259,
 * which a shard does not carry).  Ignored when the bytes start with a container header of their own. */
int leon_vlc_open_shard(const uint8_t* data, size_t n, int32_t threads, int32_t has_alpha, leon_vlc_stream** out);
/* A stream for leon_vlc_scan_picture only (the pipeline's gpu_parser mode: the host reads the layers above the slices,
 * a few dozen bytes per picture, and finds the slice start codes): when `readable` >= n + 16 bytes may be read from
 * `data` (what lies behind the n bytes does not matter), NOTHING is copied and no read-ahead thread is started -- `data`
 * must then stay valid until leon_vlc_close; leon_vlc_next_picture* are refused.  Opening and closing a 1.2 MB GOP shard
 * the ordinary way (copy, thread) cost more than scanning its twelve pictures. */
int leon_vlc_open_scan(const uint8_t* data, size_t n, size_t readable, int32_t has_alpha, leon_vlc_stream** out);
void leon_vlc_close(leon_vlc_stream* s);
int leon_vlc_get_info(leon_vlc_stream* s, leon_vlc_info* out);

/* = decodeFrame (decoders/jsv.js:426-469) without the reconstruction: hands out the next picture.
 * The stream parses one picture AHEAD on its own coordinator thread (two result sets), so the call
 * usually returns a picture that is already there while the following one is being parsed -- the
 * caller's submit and the parse overlap.  The arrays of *out belong to the stream and stay valid
 * until the next call; leon_vlc_get_info reports the sequence state as of the last picture returned. */
int leon_vlc_next_picture(leon_vlc_stream* s, leon_vlc_picture* out);

/* The same without the parse-ahead: the picture is parsed on the calling thread, during the call (what a caller
 * that runs one stream per thread anyway wants: the pipeline's GOP-shard parsers).  The arrays of *out stay valid
 * until the next call.  Not to be mixed with leon_vlc_next_picture on one stream. */
int leon_vlc_next_picture_sync(leon_vlc_stream* s, leon_vlc_picture* out);

/* The picture layer only (decodePicture's header part, decoders/jsv.js:583-650, and the slice start codes
 * :651-660): the next picture's header fields and where its slices start, nothing below the slice start code is
 * read.  For a front end that decodes the slices elsewhere -- the GPU parser of the pipeline
 * (leon_pipeline_config.gpu_parser).  Positions are relative to the bytes given to leon_vlc_open*; the arrays
 * stay valid until the next call.  Not to be mixed with leon_vlc_next_picture* on one stream. */
typedef struct leon_vlc_picture_scan {
    int32_t type, temporal_reference;
    double  ts_ms;
    int32_t new_sequence;
    int32_t full_pel_fwd, fwd_rsize;     /* forward_f = 1 << fwd_rsize (jsv.js:607-613) */
    int32_t full_pel_bwd, bwd_rsize;
    uint32_t n_slices;
    const int32_t* slice_code;           /* [n_slices] slice_vertical_position, 1 .. 175 */
    const uint64_t* slice_bit_pos;       /* [n_slices] first bit behind the slice start code */
    uint64_t end_byte;                   /* the first byte behind the picture's last slice (next start code, or the end) */
    int32_t open_gop;                    /* as in leon_vlc_picture */
    int32_t reserved;
} leon_vlc_picture_scan;
int leon_vlc_scan_picture(leon_vlc_stream* s, leon_vlc_picture_scan* out);

/* The front end's decoding tables as plain arrays (entries as in leon_vlc.cpp's Tables: (length << 16) | value,
 * 0 = invalid code), for a slice decoder that runs elsewhere: mpeg1video-decoder-webgl_amd/csrc/leon_vlc_gpu.h. */
typedef struct leon_vlc_gpu_tables {
    uint32_t fast12[4096];       /* a coefficient symbol (not a block's first) from its next 12 bits: bits 0..6 length,
                                    bit 7 end of block, bits 8..15 run, bits 16..31 level; 0 = longer code or escape */
    int32_t  coef16[65536];      /* any coefficient code without its sign bit, by the next 16 bits */
    int32_t  motion_s[2048];     /* motion_code with its sign: (len << 16) | (code + 16) */
    int32_t  mba[2048];          /* macroblock_address_increment; 34 = stuffing, 35 = escape */
    int32_t  mbtype[4][64];      /* [picture type][next 6 bits] -> flags 0x10 quant | 0x08 fwd | 0x04 bwd | 0x02 pattern | 0x01 intra */
    int32_t  cbp[512];
    int32_t  dc_lum[128], dc_chr[256];
    uint16_t zz_off[64];         /* tile byte offset r*128 + c*2 of zig-zag index n */
} leon_vlc_gpu_tables;
int leon_vlc_get_gpu_tables(leon_vlc_gpu_tables* out);

/* = jsv.prototype.seek (decoders/jsv.js:1618-1648): position on the key-map entry at or before
 * `seconds`; decoding resumes at the next sequence header.  *byte_offset receives the offset. */
int leon_vlc_seek(leon_vlc_stream* s, double seconds, uint64_t* byte_offset);

/* The GOP key map of the container header (decoders/jsv.js:264-268, :282-313): absolute byte offset
 * of each GOP's sequence header and its time code.  Fills at most `capacity` entries (either array
 * may be NULL) and returns the number of entries the stream has.  A stream opened on the bytes from
 * one entry's offset up to the next one's PLUS THREE (a "GOP shard": it starts with 00 00 01 C3 and ends
 * with the 00 00 01 of what follows, without the code byte) decodes exactly that GOP -- the unit of the
 * frame-parallel partition.  The three bytes matter: the slice loop ends a slice at a start
 * code prefix or within two bytes of the end of the data (jsv.js:1710-1760); cut at the offset itself, a
 * last macroblock of two bytes or less would be taken for that end and dropped. */
int leon_vlc_get_keymap(leon_vlc_stream* s, uint32_t* byte_offsets, uint32_t* timecodes, uint32_t capacity);

/* sparse lists -> the dense int16 planes the reference uploads (planes are overwritten) */
int leon_vlc_densify(const leon_vlc_info* info, const leon_vlc_picture* pic, int16_t* y, int16_t* cb, int16_t* cr);
/* the A plane of a yuva picture (coded-luma size) */
int leon_vlc_densify_alpha(const leon_vlc_info* info, const leon_vlc_picture* pic, int16_t* a);

#ifdef __cplusplus
}
#endif
#endif
