// fetch_calib.cpp -- calibrates rocprofv3's FETCH_SIZE (and the raw TCC_EA0 read-request counters) on
// gfx950 for the load SHAPES k_recon issues, on buffers with a KNOWN unique byte count far beyond the
// 256 MiB Infinity Cache.  One kernel per shape, each reading every byte of `bytes` exactly once
// (or, for the sparse shape, one dword per 64-byte half line):
//   calib_b128      16 B per lane, contiguous          = the coefficient-row loads (buffer_load_b128)
//   calib_b96       12 B per lane, contiguous (pitch 12)= aligned 12-byte loads (buffer_load_b96)
//   calib_b96_win   12 B per lane at 8-byte pitch, rows = the reference-row windows of k_recon: the 8 lanes
//                   of a group read 68 contiguous bytes of one picture row, 8 rows per wave
//   calib_u8        1 B per lane, contiguous            = the macroblock-map loads
//   calib_b32_half  4 B per 64-byte half line           = what a partially used line costs
// Usage (GPU box): rocprofv3 --pmc FETCH_SIZE -d out -- ./fetch_calib.bin   (and other counters in
// their own passes); tools/summarize_calibration.py turns the CSVs into profiles/<tag>_fetch_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v3u __attribute__((ext_vector_type(3)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}

// every kernel folds what it read into one dword that is (never) written: the loads cannot be dropped
#define SINK(acc) if ((acc) == 0x9e3779b9u) out[0] = (acc)

__global__ __launch_bounds__(256) void calib_b128(const char* __restrict__ src, uint32_t* __restrict__ out, uint32_t chunk_bytes)
{
    // one workgroup = 4 KiB; the grid is cut in chunks below 2 GiB so that 32-bit buffer offsets do
    const char* base = src + (size_t)blockIdx.y * chunk_bytes;
    const uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 16u;
    v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc(base), (int)off, 0, 0);
    uint32_t a = r.x ^ r.y ^ r.z ^ r.w;
    SINK(a);
}
__global__ __launch_bounds__(256) void calib_b96(const char* __restrict__ src, uint32_t* __restrict__ out, uint32_t chunk_bytes)
{
    const char* base = src + (size_t)blockIdx.y * chunk_bytes;
    const uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 12u;
    v3u r = __builtin_amdgcn_raw_buffer_load_b96(rsrc(base), (int)off, 0, 0);
    uint32_t a = r.x ^ r.y ^ r.z;
    SINK(a);
}
// planes of W x H bytes; a wave = 8 rows x 8 blocks of 8 samples: lane (n = lane>>3, b = lane&7) reads
// 12 bytes at row (8R + n), column 64g + 8b + shift, shift in {0, 4} alternating per macroblock pair
// like half of the half-pel windows do (the window start is dword aligned)
__global__ __launch_bounds__(256) void calib_b96_win(const char* __restrict__ src, uint32_t* __restrict__ out, int W, int H, uint32_t plane_bytes)
{
    const char* base = src + (size_t)blockIdx.y * plane_bytes;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int gpr = W / 64;
    const int R = wave / gpr, g = wave - R * gpr;
    if (R >= H / 8) return;
    const int n = lane >> 3, b = lane & 7;
    const uint32_t shift = ((g * 4 + (b >> 1)) & 1) ? 4u : 0u;
    const uint32_t off = (uint32_t)(8 * R + n) * (uint32_t)W + (uint32_t)(64 * g + 8 * b) + shift;
    v3u r = __builtin_amdgcn_raw_buffer_load_b96(rsrc(base), (int)off, 0, 0);
    uint32_t a = r.x ^ r.y ^ r.z;
    SINK(a);
}
__global__ __launch_bounds__(256) void calib_u8(const char* __restrict__ src, uint32_t* __restrict__ out, uint32_t chunk_bytes)
{
    const char* base = src + (size_t)blockIdx.y * chunk_bytes;
    const uint32_t off = blockIdx.x * 256u + threadIdx.x;
    uint32_t a = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc(base), (int)off, 0, 0);
    asm volatile("" : "+v"(a));          // the value range of a byte would let the compiler drop load and sink
    SINK(a);
}
__global__ __launch_bounds__(256) void calib_b32_half(const char* __restrict__ src, uint32_t* __restrict__ out, uint32_t chunk_bytes)
{
    const char* base = src + (size_t)blockIdx.y * chunk_bytes;
    const uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 64u;
    uint32_t a = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc(base), (int)off, 0, 0);
    SINK(a);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)3 << 30;                 // 3 GiB source: 12x the Infinity Cache
    char* src = nullptr;
    uint32_t* out = nullptr;
    CK(hipMalloc(&src, bytes + 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 0x5a, bytes + 4096));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const uint32_t chunk = 1u << 30;                      // 1 GiB chunks, grid.y = 3
    struct Run { const char* name; double unique_bytes; } runs[5];
    float ms[5];
    for (int rep = 0; rep < 2; rep++) {                   // second pass is the one to read (first warms clocks)
        int k = 0;
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(calib_b128, dim3(chunk / 4096, 3), dim3(256), 0, 0, src, out, chunk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[k], a, b));
        runs[k++] = {"calib_b128", (double)bytes};
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(calib_b96, dim3(chunk / 3072, 3), dim3(256), 0, 0, src, out, chunk);   // 256 lanes x 12 B = 3072 B
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[k], a, b));
        runs[k++] = {"calib_b96", (double)(chunk / 3072) * 3072.0 * 3};
        {
            const int W = 1920, H = 1088;
            const uint32_t plane = (uint32_t)W * H;
            const int planes = (int)(bytes / plane);      // 1542 planes of 2 MB
            const int waves = (W / 64) * (H / 8);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(calib_b96_win, dim3((waves + 3) / 4, planes), dim3(256), 0, 0, src, out, W, H, plane);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[k], a, b));
            runs[k++] = {"calib_b96_win", (double)plane * planes};
        }
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(calib_u8, dim3(chunk / 256, 3), dim3(256), 0, 0, src, out, chunk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[k], a, b));
        runs[k++] = {"calib_u8", (double)bytes};
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(calib_b32_half, dim3(chunk / (256 * 64), 3), dim3(256), 0, 0, src, out, chunk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[k], a, b));
        runs[k++] = {"calib_b32_half", (double)bytes};    // every 64-byte half line is touched
    }
    CK(hipGetLastError());
    printf("{\"source_bytes\": %zu, \"kernels\": {", bytes);
    for (int k = 0; k < 5; k++)
        printf("%s\"%s\": {\"unique_bytes\": %.0f, \"ms\": %.4f, \"gbps\": %.1f}", k ? ", " : "", runs[k].name, runs[k].unique_bytes, ms[k],
               runs[k].unique_bytes / (ms[k] * 1e-3) / 1e9);
    printf("}}\n");
    hipFree(src);
    hipFree(out);
    return 0;
}
