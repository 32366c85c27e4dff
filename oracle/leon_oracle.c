/*
 * leon_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * See leon_oracle.h for the scope statement and the reference citations.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, no fast-math)
 */
#include "leon_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* decoders/jsv.js:1777-1786 (= ISO 11172-2 default intra matrix) */
const uint8_t LO_DEFAULT_INTRA_QUANT[64] = {
     8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};
/* decoders/jsv.js:1787-1796 */
const uint8_t LO_DEFAULT_NON_INTRA_QUANT[64] = {
    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16,
    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16,
    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16,
    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
/* decoders/jsv.js:1797-1806 */
const uint8_t LO_PREMULTIPLIER[64] = {
    32, 44, 42, 38, 32, 25, 17,  9, 44, 62, 58, 52, 44, 35, 24, 12,
    42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
    32, 44, 42, 38, 32, 25, 17,  9, 25, 35, 33, 30, 25, 20, 14,  7,
    17, 24, 23, 20, 17, 14,  9,  5,  9, 12, 12, 10,  9,  7,  5,  2};

/* mpeg1video.js:23 (COL_INT_5) and :26 (ROWSCOM_INT4): same text in both passes.
 * '/' is GLSL int division = C truncation (decision D1). */
void lo_butterfly8(const int32_t X[8], int32_t o[8])
{
    int32_t b1, b3, b4, b6, b7, tmp1, tmp2, m0, x0, x1, x2, x3, x4, y3, y4, y5, y6, y7;
    b1 = X[4];
    b3 = X[2] + X[6];
    b4 = X[5] - X[3];
    tmp1 = X[1] + X[7];
    tmp2 = X[3] + X[5];
    b6 = X[1] - X[7];
    b7 = tmp1 + tmp2;
    m0 = X[0];
    x4 = ((b6 * 473 - b4 * 196 + 128) / 256) - b7;
    x0 = x4 - (((tmp1 - tmp2) * 362 + 128) / 256);
    x1 = m0 - b1;
    x2 = (((X[2] - X[6]) * 362 + 128) / 256) - b3;
    x3 = m0 + b1;
    y3 = x1 + x2;
    y4 = x3 + b3;
    y5 = x1 - x2;
    y6 = x3 - b3;
    y7 = -x0 - ((b4 * 473 + b6 * 196 + 128) / 256);
    o[0] = b7 + y4;
    o[1] = x4 + y3;
    o[2] = y5 - x0;
    o[3] = y6 - y7;
    o[4] = y6 + y7;
    o[5] = x0 + y5;
    o[6] = y3 - x4;
    o[7] = y4 - b7;
}

/* _B() then the RGBA8 render-target store then _E() (mpeg1video.js:18):
 *   if (w < 0) w += 65536; hi = floor(w/256); lo = w - hi*256;
 *   bytes (lo/255, hi/255) are UNORM8-stored, i.e. each clamped to [0,255];
 *   read back as two's complement int16.  For |w| < 65536 this is w mod 2^16. */
int32_t lo_handoff_store(int32_t w)
{
    int32_t v = w < 0 ? w + 65536 : w;
    int32_t hi = (int32_t)floor((double)v / 256.0);
    int32_t lo = v - hi * 256;
    if (hi < 0) hi = 0;
    if (hi > 255) hi = 255;
    if (lo < 0) lo = 0;     /* cannot happen: lo is in [0,255] by construction */
    if (lo > 255) lo = 255;
    int32_t u = hi * 256 + lo;
    return u >= 32768 ? u - 65536 : u;
}

static inline int mb_index(int is_chroma, int Q, int R, int mbw)
{
    /* pass 1 looks the per-MB maps up at block (Q,R) of the plane: luma has two
     * blocks per macroblock in each direction, chroma one (mpeg1video.js:21,
     * _h = 1/(W/8) over an mbw-wide texture; decision D8 for the edge samples). */
    return is_chroma ? R * mbw + Q : (R >> 1) * mbw + (Q >> 1);
}

/* one column (horizontal frequency z) of one block: mpeg1video.js:20-24 */
static void pass1_column(const int16_t* coef, int W, int Q, int R, int z,
                         int ag, int q, const uint8_t* qm, const uint8_t* pm, int32_t v[8])
{
    int32_t X[8];
    for (int i = 0; i < 8; i++)                          /* COL_INT_2 */
        X[i] = coef[(8 * R + i) * W + 8 * Q + z];
    int32_t dc = X[0];                                    /* COL_INT_21 */
    for (int i = 0; i < 8; i++) {                         /* COL_31 .. COL_INT_3 */
        if (X[i] == 0)
            continue;   /* "_U + 1. > last_non_zero" is always true: zeros are skipped */
        int64_t x = 2 * (int64_t)X[i];
        int O = qm[(ag ? 0 : 64) + i * 8 + z];
        if (ag == 0)
            x += x < 0 ? -1 : 1;
        /* floor(x*q*O/16.0): arithmetic shift = floor (decision D3) */
        int64_t p = x * q * O;
        int64_t f = p >= 0 ? p / 16 : -((-p + 15) / 16);
        if ((f & 1) == 0)                                 /* mod(X,2.) == 0. */
            f -= (f > 0) ? 1 : -1;                        /* note: f == 0 becomes +1 */
        if (f > 2047) f = 2047;
        if (f < -2048) f = -2048;
        X[i] = (int32_t)f * pm[i * 8 + z];
    }
    if (z == 0 && ag == 1)                                /* COL_4 + COL_INT_31 */
        X[0] = dc * 256;
    lo_butterfly8(X, v);
}

void lo_pass1_plane(const int16_t* coef, int W, int H, int is_chroma,
                    const uint8_t* qscale, const uint8_t* intra, int mbw,
                    const uint8_t* qm, const uint8_t* pm, int16_t* scratch)
{
    const float y04 = 0.4f;                               /* _y, COLUMNS_2 */
    for (int R = 0; R < H / 8; R++)
        for (int Q = 0; Q < W / 8; Q++) {
            int mb = mb_index(is_chroma, Q, R, mbw);
            int ag = intra[mb] > 0 ? 1 : 0;               /* COL_3: .r > 0. */
            int q = qscale[mb];
            for (int z = 0; z < 8; z++) {
                int32_t v[8];
                pass1_column(coef, W, Q, R, z, ag, q, qm, pm, v);
                for (int n = 0; n < 8; n++) {
                    /* _B( floor( float(v) * _y ) ): binary32 multiply, then floor */
                    float fw = floorf((float)v[n] * y04);
                    int32_t w = (int32_t)fw;
                    /* fragment row z of the block, output column n: transposed */
                    scratch[(8 * R + z) * W + 8 * Q + n] = (int16_t)lo_handoff_store(w);
                }
            }
        }
}

void lo_pass2_residual_plane(const int16_t* scratch, int W, int H, int32_t* res)
{
    const float y04 = 0.4f;
    for (int R = 0; R < H / 8; R++)
        for (int Q = 0; Q < W / 8; Q++)
            for (int a = 0; a < 8; a++) {
                int32_t X[8], t[8];
                for (int i = 0; i < 8; i++) {
                    /* ROWS_INT1/2: int( _E(texel) / _y ): binary32 divide, truncate */
                    float f = (float)scratch[(8 * R + i) * W + 8 * Q + a] / y04;
                    X[i] = (int32_t)f;
                }
                lo_butterfly8(X, t);
                for (int m = 0; m < 8; m++)
                    res[(8 * R + a) * W + 8 * Q + m] = (t[m] + 128) / 256;
            }
}

static inline uint8_t clamp_u8(int32_t v) { return v < 0 ? 0 : v > 255 ? 255 : (uint8_t)v; }

void lo_pass2_intra_plane(const int16_t* scratch, int W, int H, uint8_t* out)
{
    int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * H);
    lo_pass2_residual_plane(scratch, W, H, res);
    for (int i = 0; i < W * H; i++)
        out[i] = clamp_u8(res[i]);                        /* UNORM8 store clamps */
    free(res);
}

/* reference texel fetch: a pixel column x lives in RGBA texel x>>2, component
 * x&3, and the texel index (not the pixel) is CLAMP_TO_EDGE'd (jsv.js:216-217,
 * _p() mpeg1video.js:24).  Rows clamp per pixel row. */
static inline uint8_t ref_px(const uint8_t* ref, int W, int H, int x, int y)
{
    int t = x >> 2;                 /* floor for negatives */
    int c = x & 3;
    if (t < 0) t = 0;
    if (t > W / 4 - 1) t = W / 4 - 1;
    if (y < 0) y = 0;
    if (y > H - 1) y = H - 1;
    return ref[y * W + 4 * t + c];
}

static inline int trunc_half(int v) { return v / 2; }    /* _F(v/2.): toward zero */
static inline int floor_half(int v) { return v >> 1; }   /* floor(v/2.) */

void lo_predict_plane(const uint8_t* ref, int W, int H, int is_chroma,
                      const int16_t* mv, int mbw, uint8_t* pred)
{
    int mbs = is_chroma ? 8 : 16;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int mb = (y / mbs) * mbw + (x / mbs);
            int mh = mv[2 * mb], mvv = mv[2 * mb + 1];
            int ax, ay, oh, ov;
            if (!is_chroma) {                  /* _ac == 1. (mpeg1video.js:28) */
                ax = floor_half(mh);  ay = floor_half(mvv);
                oh = mh & 1;          ov = mvv & 1;
            } else {                           /* chroma: trunc, then floor */
                int h = trunc_half(mh), v = trunc_half(mvv);
                ax = floor_half(h);   ay = floor_half(v);
                oh = h & 1;           ov = v & 1;
            }
            int a = ref_px(ref, W, H, x + ax, y + ay);
            int p;
            if (oh && ov)
                p = (a + ref_px(ref, W, H, x + ax + 1, y + ay) + ref_px(ref, W, H, x + ax, y + ay + 1) +
                     ref_px(ref, W, H, x + ax + 1, y + ay + 1) + 2) >> 2;
            else if (oh)
                p = (a + ref_px(ref, W, H, x + ax + 1, y + ay) + 1) >> 1;
            else if (ov)
                p = (a + ref_px(ref, W, H, x + ax, y + ay + 1) + 1) >> 1;
            else
                p = a;
            pred[y * W + x] = (uint8_t)p;
        }
}

void lo_pass2_inter_plane(const int16_t* scratch, int W, int H, int is_chroma,
                          const uint8_t* repadd, const int16_t* mv, int mbw,
                          const uint8_t* ref, uint8_t* out)
{
    int mbs = is_chroma ? 8 : 16;
    int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * H);
    uint8_t* pred = (uint8_t*)malloc((size_t)W * H);
    lo_pass2_residual_plane(scratch, W, H, res);
    lo_predict_plane(ref, W, H, is_chroma, mv, mbw, pred);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int mb = (y / mbs) * mbw + (x / mbs);
            int p = repadd[mb] >= 128 ? 0 : pred[y * W + x];   /* .r > 0.5 */
            out[y * W + x] = clamp_u8(res[y * W + x] + p);
        }
    free(res);
    free(pred);
}

void lo_pass2_bidir_plane(const int16_t* scratch, int W, int H, int is_chroma,
                          const uint8_t* repadd, const uint8_t* mb_dir,
                          const int16_t* mv_fwd, const int16_t* mv_bwd, int mbw,
                          const uint8_t* ref_fwd, const uint8_t* ref_bwd, uint8_t* out)
{
    int mbs = is_chroma ? 8 : 16;
    int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * H);
    uint8_t* pf = (uint8_t*)malloc((size_t)W * H);
    uint8_t* pb = (uint8_t*)malloc((size_t)W * H);
    lo_pass2_residual_plane(scratch, W, H, res);
    lo_predict_plane(ref_fwd, W, H, is_chroma, mv_fwd, mbw, pf);
    lo_predict_plane(ref_bwd, W, H, is_chroma, mv_bwd, mbw, pb);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int mb = (y / mbs) * mbw + (x / mbs);
            int i = y * W + x, p;
            int d = mb_dir[mb] & 3;
            if (repadd[mb] >= 128 || d == 0) p = 0;
            else if (d == 1) p = pf[i];
            else if (d == 2) p = pb[i];
            else p = (pf[i] + pb[i] + 1) >> 1;
            out[i] = clamp_u8(res[i] + p);
        }
    free(res);
    free(pf);
    free(pb);
}

void lo_decode_picture(int type, int cw, int ch,
                       const int16_t* coef_y, const int16_t* coef_cb, const int16_t* coef_cr,
                       const uint8_t* qscale, const uint8_t* intra, const uint8_t* repadd,
                       const uint8_t* mb_dir, const int16_t* mv_fwd, const int16_t* mv_bwd,
                       const uint8_t* qm, const uint8_t* pm,
                       const uint8_t* ref_fwd, const uint8_t* ref_bwd, uint8_t* out)
{
    int mbw = cw / 16;
    const int16_t* coef[3] = {coef_y, coef_cb, coef_cr};
    size_t off = 0;
    for (int comp = 0; comp < 3; comp++) {
        int W = comp ? cw / 2 : cw, H = comp ? ch / 2 : ch, isc = comp != 0;
        int16_t* scratch = (int16_t*)malloc(sizeof(int16_t) * (size_t)W * H);
        lo_pass1_plane(coef[comp], W, H, isc, qscale, intra, mbw, qm, pm, scratch);
        if (type == LO_PIC_I)
            lo_pass2_intra_plane(scratch, W, H, out + off);
        else if (type == LO_PIC_P)
            lo_pass2_inter_plane(scratch, W, H, isc, repadd, mv_fwd, mbw, ref_fwd + off, out + off);
        else
            lo_pass2_bidir_plane(scratch, W, H, isc, repadd, mb_dir, mv_fwd, mv_bwd, mbw,
                                 ref_fwd + off, ref_bwd + off, out + off);
        free(scratch);
        off += (size_t)W * H;
    }
}

/* Uint8ClampedArray element store: clamp to [0,255], round half to even */
static inline uint8_t to_u8_clamped(double v)
{
    if (!(v > 0.0)) return 0;       /* also NaN */
    if (v > 255.0) return 255;
    return (uint8_t)lrint(v);        /* default rounding mode: nearest-even */
}

/* player/easybits.player.js:2674-2785.  The index progression is restated
 * literally: for an odd frame width the reference's yNext2Lines / rgbaNext2Lines
 * bookkeeping drifts by one sample per row pair, and so does this. */
void lo_ycbcr_to_rgba_cpu(const uint8_t* pY, const uint8_t* pCb, const uint8_t* pCr,
                          int coded_w, int frame_w, int frame_h, uint8_t* rgba)
{
    int half_w = coded_w >> 1;
    memset(rgba, 255, (size_t)frame_w * frame_h * 4);      /* fillArray(pRGBA, 255) */
    int yIndex1 = 0, yIndex2 = coded_w;
    int yNext2Lines = coded_w + (coded_w - frame_w);
    int cIndex = 0, cNextLine = half_w - (frame_w >> 1);
    int rgbaIndex1 = 0, rgbaIndex2 = frame_w * 4, rgbaNext2Lines = frame_w * 4;
    int cols = frame_w >> 1, rows = frame_h >> 1;
    for (int row = 0; row < rows; row++) {
        for (int col = 0; col < cols; col++) {
            double cb = pCb[cIndex], cr = pCr[cIndex];
            cIndex++;
            double yuvr = cr - 128, yuvb = cb - 128;
            double r = yuvr * 1.59603;
            double g1 = -0.81297 * yuvr;
            double g2 = 0.39176 * yuvb;
            double g = g1 - g2;
            double b = yuvb * 2.01723;
            for (int k = 0; k < 2; k++) {                   /* line 1: two pixels */
                double ys = ((double)pY[yIndex1++] - 16) * 1.16438;
                rgba[rgbaIndex1] = to_u8_clamped(r + ys);
                rgba[rgbaIndex1 + 1] = to_u8_clamped(g + ys);
                rgba[rgbaIndex1 + 2] = to_u8_clamped(b + ys);
                rgbaIndex1 += 4;
            }
            for (int k = 0; k < 2; k++) {                   /* line 2 */
                double ys = ((double)pY[yIndex2++] - 16) * 1.16438;
                rgba[rgbaIndex2] = to_u8_clamped(r + ys);
                rgba[rgbaIndex2 + 1] = to_u8_clamped(g + ys);
                rgba[rgbaIndex2 + 2] = to_u8_clamped(b + ys);
                rgbaIndex2 += 4;
            }
        }
        yIndex1 += yNext2Lines;
        yIndex2 += yNext2Lines;
        rgbaIndex1 += rgbaNext2Lines;
        rgbaIndex2 += rgbaNext2Lines;
        cIndex += cNextLine;
    }
}

/* player/parts/end.js:77-156: texel/255 -> vec4 * mat4 in binary32 -> UNORM8.
 * Evaluation order inside the dot products is implementation-defined in GLSL;
 * this restatement fixes it left to right without contraction ("by fiat").
 * The GL path converts every pixel of the frame_w x frame_h crop. */
void lo_ycbcr_to_rgba_gl(const uint8_t* pY, const uint8_t* pCb, const uint8_t* pCr,
                         int coded_w, int frame_w, int frame_h, uint8_t* rgba)
{
    int half_w = coded_w >> 1;
    static const float M[3][4] = {{1.16438f, 0.00000f, 1.59603f, -0.87079f},
                                  {1.16438f, -0.39176f, -0.81297f, 0.52959f},
                                  {1.16438f, 2.01723f, 0.00000f, -1.08139f}};
    for (int py = 0; py < frame_h; py++)
        for (int px = 0; px < frame_w; px++) {
            float fy = (float)pY[py * coded_w + px] / 255.0f;
            float fcb = (float)pCb[(py >> 1) * half_w + (px >> 1)] / 255.0f;
            float fcr = (float)pCr[(py >> 1) * half_w + (px >> 1)] / 255.0f;
            uint8_t* o = rgba + ((size_t)py * frame_w + px) * 4;
            for (int c = 0; c < 3; c++) {
                float t0 = fy * M[c][0];
                float t1 = fcb * M[c][1];
                float t2 = fcr * M[c][2];
                float s = t0 + t1;
                s = s + t2;
                s = s + M[c][3];
                float cl = s < 0.0f ? 0.0f : s > 1.0f ? 1.0f : s;
                o[c] = (uint8_t)lrintf(cl * 255.0f);
            }
            o[3] = 255;
        }
}
