// VALU issue-rate probe (ad-hoc): cycles per wave64 instruction per SIMD for several opcodes
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
#define BODY16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define KERNEL(NAME, DECL, OPM, SINK)                                                    \
    __global__ __launch_bounds__(256) void NAME(int* out, int seed)                        \
    {                                                                                     \
        DECL;                                                                             \
        for (int it = 0; it < ITER; it++) { BODY16(OPM) }                                 \
        SINK;                                                                             \
    }
#define DECL_I int r[16]; for (int k = 0; k < 16; k++) r[k] = seed + k + threadIdx.x; int c = seed | 3
#define SINK_I int s = 0; for (int k = 0; k < 16; k++) s ^= r[k]; if (s == 0x7fffffff) out[0] = s
#define DECL_F float r[16]; for (int k = 0; k < 16; k++) r[k] = (float)(seed + k + threadIdx.x); float c = (float)seed * 1.0001f
#define SINK_F float s = 0; for (int k = 0; k < 16; k++) s += r[k]; if (s == 12345.0f) out[0] = (int)s
typedef float v2f __attribute__((ext_vector_type(2)));
#define DECL_F2 v2f r[16]; for (int k = 0; k < 16; k++) r[k] = v2f{(float)(seed + k + threadIdx.x), (float)k}; v2f c = {(float)seed * 1.0001f, 1.5f}
#define SINK_F2 v2f s = {0, 0}; for (int k = 0; k < 16; k++) s += r[k]; if (s.x == 12345.0f) out[0] = (int)s.y

#define OP_ADD(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_MAD24(k) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[k]) : "v"(c));
#define OP_ASHR(k) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r[k]));
#define OP_MED3(k) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_LERP(k) asm volatile("v_lerp_u8 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_ADDF(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_FMAF(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_CVT(k) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[k]));
#define OP_FLOOR(k) asm volatile("v_floor_f32 %0, %0" : "+v"(r[k]));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_PKADDI16(k) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_ASHRPK(k) asm volatile("v_ashr_pk_u8_i32 %0, %0, %1, 8" : "+v"(r[k]) : "v"(c));

#define OP_MUL24(k) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_MULU24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(c));
#define OP_ALIGNBYTE(k) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r[k]) : "v"(c));
#define OP_BFE(k) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(r[k]));
#define OP_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_LSHLOR(k) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(r[k]) : "v"(c));
#define OP_MAX(k) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_LSHL(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[k]));
#define OP_BITOP3(k) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0xde" : "+v"(r[k]) : "v"(c));
#define OP_SATPK(k) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(r[k]));
#define OP_ADDSDWA(k) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(r[k]) : "v"(c));
#define OP_MULSDWA(k) asm volatile("v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[k]) : "v"(c));
#define OP_CMP(k) asm volatile("v_cmp_lt_i32 vcc, %0, %1" :: "v"(r[k]), "v"(c) : "vcc");
#define OP_CVTFLR(k) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(r[k]));
#define OP_CVTI(k) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[k]));
#define OP_MULF(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_FMAC(k) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_MAX3F(k) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_TRUNC(k) asm volatile("v_trunc_f32 %0, %0" : "+v"(r[k]));
#define OP_CVTUB(k) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[k]));
#define OP_CVTPKU8(k) asm volatile("v_cvt_pk_u8_f32 %0, %0, 1, %1" : "+v"(r[k]) : "v"(c));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_MOVDPP(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[k]) : "v"(c));
#define OP_ADDDPP(k) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[k]) : "v"(c));
typedef double dbl;
#define DECL_D double r[16]; for (int k = 0; k < 16; k++) r[k] = (double)(seed + k + threadIdx.x); double c = (double)seed * 1.0001
#define SINK_D double s = 0; for (int k = 0; k < 16; k++) s += r[k]; if (s == 12345.0) out[0] = (int)s
#define OP_ADDD(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_MULD(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define OP_FMAD(k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
#define OP_CVTDU(k) { unsigned u_ = (unsigned)k + threadIdx.x; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r[k]) : "v"(u_)); }
#define OP_CVTID(k) { int i_; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i_) : "v"(r[k])); asm volatile("" :: "v"(i_)); }
#define OP_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r[k]) : "v"(c));
#define OP_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(r[k]) : "v"(c));
KERNEL(k_addd, DECL_D, OP_ADDD, SINK_D)
KERNEL(k_muld, DECL_D, OP_MULD, SINK_D)
KERNEL(k_fmad, DECL_D, OP_FMAD, SINK_D)
KERNEL(k_cvtdu, DECL_D, OP_CVTDU, SINK_D)
KERNEL(k_cvtid, DECL_D, OP_CVTID, SINK_D)
KERNEL(k_bperm, DECL_I, OP_BPERM, SINK_I)
KERNEL(k_mov, DECL_I, OP_MOV, SINK_I)
KERNEL(k_mul24, DECL_I, OP_MUL24, SINK_I)
KERNEL(k_mulu24, DECL_I, OP_MULU24, SINK_I)
KERNEL(k_mulhi, DECL_I, OP_MULHI, SINK_I)
KERNEL(k_cndmask, DECL_I, OP_CNDMASK, SINK_I)
KERNEL(k_alignbyte, DECL_I, OP_ALIGNBYTE, SINK_I)
KERNEL(k_bfe, DECL_I, OP_BFE, SINK_I)
KERNEL(k_andor, DECL_I, OP_ANDOR, SINK_I)
KERNEL(k_add3, DECL_I, OP_ADD3, SINK_I)
KERNEL(k_lshlor, DECL_I, OP_LSHLOR, SINK_I)
KERNEL(k_max, DECL_I, OP_MAX, SINK_I)
KERNEL(k_and, DECL_I, OP_AND, SINK_I)
KERNEL(k_xor, DECL_I, OP_XOR, SINK_I)
KERNEL(k_lshl, DECL_I, OP_LSHL, SINK_I)
KERNEL(k_bitop3, DECL_I, OP_BITOP3, SINK_I)
KERNEL(k_satpk, DECL_I, OP_SATPK, SINK_I)
KERNEL(k_addsdwa, DECL_I, OP_ADDSDWA, SINK_I)
KERNEL(k_mulsdwa, DECL_I, OP_MULSDWA, SINK_I)
KERNEL(k_cmp, DECL_I, OP_CMP, SINK_I)
KERNEL(k_movdpp, DECL_I, OP_MOVDPP, SINK_I)
KERNEL(k_adddpp, DECL_I, OP_ADDDPP, SINK_I)
KERNEL(k_cvtflr, DECL_F, OP_CVTFLR, SINK_F)
KERNEL(k_cvti, DECL_F, OP_CVTI, SINK_F)
KERNEL(k_mulf, DECL_F, OP_MULF, SINK_F)
KERNEL(k_fmac, DECL_F, OP_FMAC, SINK_F)
KERNEL(k_max3f, DECL_F, OP_MAX3F, SINK_F)
KERNEL(k_trunc, DECL_F, OP_TRUNC, SINK_F)
KERNEL(k_cvtub, DECL_F, OP_CVTUB, SINK_F)
KERNEL(k_cvtpku8, DECL_F, OP_CVTPKU8, SINK_F)
KERNEL(k_pkmul, DECL_F2, OP_PKMUL, SINK_F2)
KERNEL(k_add, DECL_I, OP_ADD, SINK_I)
KERNEL(k_mad24, DECL_I, OP_MAD24, SINK_I)
KERNEL(k_lshladd, DECL_I, OP_LSHLADD, SINK_I)
KERNEL(k_ashr, DECL_I, OP_ASHR, SINK_I)
KERNEL(k_med3, DECL_I, OP_MED3, SINK_I)
KERNEL(k_perm, DECL_I, OP_PERM, SINK_I)
KERNEL(k_lerp, DECL_I, OP_LERP, SINK_I)
KERNEL(k_mullo, DECL_I, OP_MULLO, SINK_I)
KERNEL(k_pkaddi16, DECL_I, OP_PKADDI16, SINK_I)
KERNEL(k_ashrpk, DECL_I, OP_ASHRPK, SINK_I)
KERNEL(k_addf, DECL_F, OP_ADDF, SINK_F)
KERNEL(k_fmaf, DECL_F, OP_FMAF, SINK_F)
KERNEL(k_cvt, DECL_F, OP_CVT, SINK_F)
KERNEL(k_floor, DECL_F, OP_FLOOR, SINK_F)
KERNEL(k_pkadd, DECL_F2, OP_PKADD, SINK_F2)
KERNEL(k_pkfma, DECL_F2, OP_PKFMA, SINK_F2)

int main()
{
    int* d; hipMalloc(&d, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount; double mhz = p.clockRate / 1e3;
    printf("CUs %d clock %.0f MHz\n", cus, mhz);
    auto run = [&](const char* name, void (*k)(int*, int), int wgs_per_cu) {
        int grid = cus * wgs_per_cu;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 1); hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 1); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        double waves_per_simd = wgs_per_cu * 4 / 4.0;   // 4 waves per WG over 4 SIMDs
        double instr_per_simd = waves_per_simd * ITER * 16.0;
        double us = best * 1e3;
        printf("%-12s %.2f ns/instr/SIMD  = %.2f wave64 instr per CU per clock at %.0f MHz\n", name, us * 1e3 / instr_per_simd, 4.0 / (us * 1e3 / instr_per_simd * mhz * 1e-3), mhz);
    };
#define RUN(K) run(#K, K, 8);
    RUN(k_add) RUN(k_mad24) RUN(k_lshladd) RUN(k_ashr) RUN(k_med3) RUN(k_perm) RUN(k_lerp) RUN(k_mullo) RUN(k_pkaddi16) RUN(k_ashrpk)
    RUN(k_addf) RUN(k_fmaf) RUN(k_cvt) RUN(k_floor) RUN(k_pkadd) RUN(k_pkfma)
    RUN(k_mul24) RUN(k_mulu24) RUN(k_mulhi) RUN(k_cndmask) RUN(k_alignbyte) RUN(k_bfe) RUN(k_andor) RUN(k_add3) RUN(k_lshlor) RUN(k_max)
    RUN(k_and) RUN(k_xor) RUN(k_lshl) RUN(k_bitop3) RUN(k_satpk) RUN(k_addsdwa) RUN(k_mulsdwa) RUN(k_cmp) RUN(k_movdpp) RUN(k_adddpp)
    RUN(k_addd) RUN(k_muld) RUN(k_fmad) RUN(k_cvtdu) RUN(k_cvtid) RUN(k_bperm) RUN(k_mov)
    RUN(k_cvtflr) RUN(k_cvti) RUN(k_mulf) RUN(k_fmac) RUN(k_max3f) RUN(k_trunc) RUN(k_cvtub) RUN(k_cvtpku8) RUN(k_pkmul)
    return 0;
}
