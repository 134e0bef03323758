#!/bin/bash
# Builds timing-only variants of ld_attn_q128.hip (LD_Q128_ABLATE bits, see the file) into landiff_amd/variants/ and, on a GPU box,
# times each against the shipped kernels:  tools/attn_q128_ablate.sh build   (here, cross-compile)
#                                          tools/attn_q128_ablate.sh run     (on the GPU box)
set -e
cd "$(dirname "$0")/.."
BITS="256 257 259 260 264 287 271"
if [ "$1" = "build" ]; then
  mkdir -p landiff_amd/variants
  for b in $BITS; do
    ( cd landiff_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -fno-slp-vectorize -DLD_Q128_ABLATE=$b \
        -c ld_attn_q128.hip -o /tmp/q128_abl$b.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libq128_abl$b.so $(ls obj/*.o | grep -v ld_attn_q128) /tmp/q128_abl$b.o ) &
  done
  wait
  ls landiff_amd/variants
else
  LD_ATTN_Q128=1 python tools/attn_time.py
  for b in $BITS; do
    echo -n "ablate $b: "; LANDIFF_HIP_LIB=$PWD/landiff_amd/variants/libq128_abl$b.so LD_ATTN_Q128=1 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids
  done
  LD_ATTN_Q128=0 python tools/attn_time.py
fi
