#!/bin/bash
# Builds the current csrc/ into build/ab/<name>/libleon_hip.so (extra hipcc flags after the name), for A/B runs
# of kernel variants in ONE gpurun call (box-to-box differences are larger than most kernel changes):
#   tools/ab_build.sh base;  <edit>;  tools/ab_build.sh try -DLEON_X=1
#   gpurun -- 'for v in base try; do LEON_DEBUG_LIB=build/ab/$v/libleon_hip.so python bench.py --no-cpu-baseline --no-second-recipe; done'
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/ab/$name
mkdir -p "$out"
cp "$root/mpeg1video-decoder-webgl_amd/lib/libleon_vlc.so" "$out/"
cd "$root/mpeg1video-decoder-webgl_amd/csrc"
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -pthread -shared \
    -o "$out/libleon_hip.so" leon_hip.cpp -L"$out" -lleon_vlc -Wl,-rpath,'$ORIGIN'
echo "$out/libleon_hip.so"
