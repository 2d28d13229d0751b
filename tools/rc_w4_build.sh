#!/bin/bash
# lab library with the one-wave-per-SIMD kernel: the product's objects + efts_resconv.hip compiled with -DRC_W4=${1:-1} -> lab/rc_w4.so
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRC_W4=${1:-1} -c efficient_tts_amd/csrc/efts_resconv.hip -o /tmp/rc4.o || exit 1
mkdir -p lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lab/rc_w4.so $(ls efficient_tts_amd/build/*.o | grep -v efts_resconv.o) /tmp/rc4.o && ls -la lab/rc_w4.so
