// vlc_fuzz.cpp -- corruption fuzz of the native bitstream front end (CPU only; build with -fsanitize=address,undefined):
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -pthread -o /tmp/vlc_fuzz tools/vlc_fuzz.cpp mpeg1video-decoder-webgl_amd/csrc/leon_vlc.cpp && /tmp/vlc_fuzz tests/golden/streams/*.jsv
// Random byte and bit damage and truncation; every case must end in a clean refusal or a clean parse.
#include "../include/leon_vlc.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int fi = 1; fi < argc; fi++) {
        FILE* f = fopen(argv[fi], "rb"); std::vector<uint8_t> d(1 << 20); size_t n = fread(d.data(), 1, d.size(), f); fclose(f); d.resize(n);
        std::mt19937 rng(1234 + fi);
        for (int it = 0; it < (getenv("LEON_FUZZ_CASES") ? atoi(getenv("LEON_FUZZ_CASES")) : 600); it++) {
            std::vector<uint8_t> c = d;
            int k = 1 + rng() % 8;
            for (int j = 0; j < k; j++) { size_t at = 40 + rng() % (n - 40); if (rng() & 1) c[at] = (uint8_t)rng(); else c[at] ^= 1u << (rng() & 7); }
            if (it % 7 == 0) c.resize(40 + rng() % (n - 40));      // truncated
            leon_vlc_stream* s;
            if (leon_vlc_open(c.data(), c.size(), 1 + it % 3, &s)) { bad++; continue; }
            leon_vlc_picture p; int rc; int pics = 0;
            while ((rc = leon_vlc_next_picture(s, &p)) == 1 && pics < 100) pics++;
            if (rc < 0) bad++; else ok++;
            leon_vlc_close(s);
        }
    }
    printf("%d parsed to the end, %d refused\n", ok, bad);
}
