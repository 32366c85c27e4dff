/*
 * leon_oracle.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the reference's per-picture macroblock reconstruction
 * (dequant -> 8x8 IDCT -> motion compensation -> YCbCr->RGBA).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (libleon_hip.so) never links, loads or falls back to it.
 *
 * What each function follows in /root/reference (file:line):
 *   lo_pass1_plane        decoders/shaders/mpeg1video.js:19-24 as composed by
 *                         decoders/jsv.js:2461-2462 (integer flavour), driven by
 *                         decoders/jsv.js:1197-1268
 *   lo_pass2_*_plane      decoders/shaders/mpeg1video.js:24-29, jsv.js:2463-2464,
 *                         driven by jsv.js:1272-1334; CPU twin of the predictor
 *                         arithmetic: jsv.prototype.copyMacroblock jsv.js:895-1129
 *   lo_ycbcr_to_rgba_cpu  player/easybits.player.js:2674-2785 (YCbCrToRGBA, fp64)
 *   lo_ycbcr_to_rgba_gl   player/parts/end.js:77-156 (+ renderFrameGL
 *                         player/easybits.player.js:2787-2858)
 * Backward / bidirectional prediction (picture type B) is NOT in the reference
 * (jsv.js:613-616 returns early); it follows ISO/IEC 11172-2 2.4.4.3 and is
 * "parity unpinned by the reference".
 *
 * Arithmetic model: GLSL int = int32 with C division,
 * float = IEEE binary32 without contraction.  Pinning: tests/golden/ holds
 * (a) outputs of the reference's copyMacroblock and YCbCrToRGBA run under Node,
 * (b) boundary tensors from the unmodified reference parser, (c) outputs of a
 * literal numpy-fp32 per-fragment emulation of the composed GLSL text
 * (tools/glsl_literal.py), against all of which this file is checked.
 */
#ifndef LEON_ORACLE_H
#define LEON_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* one 8-point butterfly, identical text in both passes (mpeg1video.js:23, :26) */
void lo_butterfly8(const int32_t x[8], int32_t out[8]);

/* the int16 hand-off between the passes: _B() on write, _E() on read
 * (mpeg1video.js:18) including the UNORM8 saturation of the high byte. */
int32_t lo_handoff_store(int32_t w);

/* pass 1 ("idct_columns") over one plane.  coef: W*H int16 raw levels;
 * qscale/intra: per-macroblock maps mbw wide; qm: 128 bytes (rows 0-7 intra,
 * 8-15 non-intra, natural order); pm: 64-byte premultiplier.
 * scratch: W*H int16, laid out as the idct_1d texture holds it (transposed
 * inside each 8x8 block), minus the plane-level vertical flip that pass 2 undoes. */
void lo_pass1_plane(const int16_t* coef, int W, int H, int is_chroma,
                    const uint8_t* qscale, const uint8_t* intra, int mbw,
                    const uint8_t* qm, const uint8_t* pm, int16_t* scratch);

/* pass 2 residual only: res[y*W+x] = (t+128)/256 before any clamp (int32) */
void lo_pass2_residual_plane(const int16_t* scratch, int W, int H, int32_t* res);

/* pass 2 "idct_rows_intra": out = clamp(res, 0, 255) */
void lo_pass2_intra_plane(const int16_t* scratch, int W, int H, uint8_t* out);

/* forward predictor for one plane with the reference's vector derivation and
 * CLAMP_TO_EDGE texel-granular addressing (jsv.js:216-217).  pred: W*H u8. */
void lo_predict_plane(const uint8_t* ref, int W, int H, int is_chroma,
                      const int16_t* mv, int mbw, uint8_t* pred);

/* pass 2 "idct_rows_inter": out = clamp(res + (repadd ? 0 : pred_fwd)) */
void lo_pass2_inter_plane(const int16_t* scratch, int W, int H, int is_chroma,
                          const uint8_t* repadd, const int16_t* mv, int mbw,
                          const uint8_t* ref, uint8_t* out);

/* B pictures (ISO 11172-2; beyond the reference).  mb_dir bit0 = forward,
 * bit1 = backward; both = (pf + pb + 1) >> 1; repadd (intra MB) wins. */
void lo_pass2_bidir_plane(const int16_t* scratch, int W, int H, int is_chroma,
                          const uint8_t* repadd, const uint8_t* mb_dir,
                          const int16_t* mv_fwd, const int16_t* mv_bwd, int mbw,
                          const uint8_t* ref_fwd, const uint8_t* ref_bwd, uint8_t* out);

#define LO_PIC_I 1
#define LO_PIC_P 2
#define LO_PIC_B 3

/* One whole picture = what jsv.prototype.IDCT_GL (jsv.js:1177-1336) does.
 * Planes are contiguous [Y | Cb | Cr] (coded size, chroma half each). */
void lo_decode_picture(int type, int coded_w, int coded_h,
                       const int16_t* coef_y, const int16_t* coef_cb, const int16_t* coef_cr,
                       const uint8_t* qscale, const uint8_t* intra, const uint8_t* repadd,
                       const uint8_t* mb_dir, const int16_t* mv_fwd, const int16_t* mv_bwd,
                       const uint8_t* qm, const uint8_t* pm,
                       const uint8_t* ref_fwd, const uint8_t* ref_bwd, uint8_t* out);

/* K3, CPU twin (fp64, Uint8ClampedArray store, 2x2 quads, alpha and any odd
 * last row/column left at 255).  rgba: frame_w*frame_h*4 bytes. */
void lo_ycbcr_to_rgba_cpu(const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                          int coded_w, int frame_w, int frame_h, uint8_t* rgba);

/* K3, GL twin (fp32 matrix, left-to-right mul/add, UNORM8 store). */
void lo_ycbcr_to_rgba_gl(const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                         int coded_w, int frame_w, int frame_h, uint8_t* rgba);

/* default tables (decoders/jsv.js:1777-1806) */
extern const uint8_t LO_DEFAULT_INTRA_QUANT[64];
extern const uint8_t LO_DEFAULT_NON_INTRA_QUANT[64];
extern const uint8_t LO_PREMULTIPLIER[64];

#ifdef __cplusplus
}
#endif
#endif
