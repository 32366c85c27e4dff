#include "../../include/leon_vlc.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
int main(int argc, char** argv) {
    FILE* f = fopen("tools/probe/stream_1080p_2gop.bin", "rb");
    std::vector<uint8_t> d(3000000); size_t n = fread(d.data(), 1, d.size(), f); fclose(f);
    int reps = argc > 1 ? atoi(argv[1]) : 10; int pics = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) {
        leon_vlc_stream* s; if (leon_vlc_open(d.data(), n, 1, &s)) return 1;
        leon_vlc_picture p; while (leon_vlc_next_picture(s, &p) == 1) pics++;
        leon_vlc_close(s);
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d pictures %.3f s -> %.1f pictures/s\n", pics, dt, pics / dt);
}
