// Host -> device copy rate of pinned memory in the pipeline's shape: many 1.2 MB pieces (one GOP of 1080p stream bytes
// each), on 1 / 2 / 4 / 8 streams, and as one piece.    hipcc -O2 -o h2d_probe.bin h2d_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main()
{
    const size_t piece = 1227264, n = 512, total = piece * n;
    char *h = nullptr, *d = nullptr;
    if (hipHostMalloc((void**)&h, total, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&d, total) != hipSuccess) return 1;
    for (size_t i = 0; i < total; i += 4096) h[i] = (char)i;
    printf("{");
    for (int ns : {1, 2, 4, 8}) {
        std::vector<hipStream_t> s(ns);
        for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t i = 0; i < n; i++) hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, s[i % ns]);
            hipDeviceSynchronize();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("\"%d streams x 1.2 MB pieces\": %.1f, ", ns, total / dt / 1e9);
        }
        for (auto& x : s) hipStreamDestroy(x);
    }
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipMemcpy(d, h, total, hipMemcpyHostToDevice);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("\"one piece of %zu MB\": %.1f, \"unit\": \"GB/s\"}\n", total >> 20, total / dt / 1e9);
    return 0;
}
