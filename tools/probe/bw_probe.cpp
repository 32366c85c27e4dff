// HBM bandwidth probe: several copy / read / write kernel shapes over 2 GiB (ad-hoc tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_gs(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n)
{ size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256; for (; i < n; i += st) d[i] = s[i]; }
template <int U> __global__ __launch_bounds__(256) void copy_blk(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n)
{ size_t b = ((size_t)blockIdx.x * U) * 256 + threadIdx.x; v4u r[U];
#pragma unroll
  for (int u = 0; u < U; u++) r[u] = s[b + (size_t)u * 256];
#pragma unroll
  for (int u = 0; u < U; u++) d[b + (size_t)u * 256] = r[u]; }
template <int U> __global__ __launch_bounds__(256) void copy_blk_nt(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n)
{ size_t b = ((size_t)blockIdx.x * U) * 256 + threadIdx.x; v4u r[U];
#pragma unroll
  for (int u = 0; u < U; u++) r[u] = __builtin_nontemporal_load(&s[b + (size_t)u * 256]);
#pragma unroll
  for (int u = 0; u < U; u++) __builtin_nontemporal_store(r[u], &d[b + (size_t)u * 256]); }
template <int U> __global__ __launch_bounds__(256) void read_blk(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n)
{ size_t b = ((size_t)blockIdx.x * U) * 256 + threadIdx.x; v4u a = {0,0,0,0};
#pragma unroll
  for (int u = 0; u < U; u++) { v4u r = s[b + (size_t)u * 256]; a ^= r; }
  if (a.x == 0x12345678 && a.y == 1) d[0] = a; }
template <int U> __global__ __launch_bounds__(256) void write_blk(v4u* __restrict__ d, size_t n)
{ size_t b = ((size_t)blockIdx.x * U) * 256 + threadIdx.x; v4u a = {1,2,3,(unsigned)b};
#pragma unroll
  for (int u = 0; u < U; u++) d[b + (size_t)u * 256] = a; }
int main()
{
    size_t bytes = (size_t)2 << 30, n = bytes / 16;
    v4u *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 0x5a, bytes); hipMemset(d, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, double moved, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-28s %8.1f GB/s  (%.3f ms)\n", name, moved / (best * 1e-3) / 1e9, best);
    };
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536})
        run(("copy grid-stride g=" + std::to_string(g)).c_str(), 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_gs, dim3(g), dim3(256), 0, 0, s, d, n); });
    run("copy blk U=1", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_blk<1>, dim3(n / 256), dim3(256), 0, 0, s, d, n); });
    run("copy blk U=2", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_blk<2>, dim3(n / 512), dim3(256), 0, 0, s, d, n); });
    run("copy blk U=4", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_blk<4>, dim3(n / 1024), dim3(256), 0, 0, s, d, n); });
    run("copy blk U=8", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_blk<8>, dim3(n / 2048), dim3(256), 0, 0, s, d, n); });
    run("copy blk nt U=4", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_blk_nt<4>, dim3(n / 1024), dim3(256), 0, 0, s, d, n); });
    run("copy blk nt U=8", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_blk_nt<8>, dim3(n / 2048), dim3(256), 0, 0, s, d, n); });
    run("read blk U=4", 1.0 * bytes, [&] { hipLaunchKernelGGL(read_blk<4>, dim3(n / 1024), dim3(256), 0, 0, s, d, n); });
    run("read blk U=8", 1.0 * bytes, [&] { hipLaunchKernelGGL(read_blk<8>, dim3(n / 2048), dim3(256), 0, 0, s, d, n); });
    run("write blk U=4", 1.0 * bytes, [&] { hipLaunchKernelGGL(write_blk<4>, dim3(n / 1024), dim3(256), 0, 0, d, n); });
    run("hipMemcpyDtoD", 2.0 * bytes, [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
