#!/bin/bash
# Builds liblandiff_hip_ablate<mask>.so variants of the DiT attention kernel (LD_ATTN_ABLATE, see ld_attn_p16.hip) next to the
# shipped library: timing experiments that show what each non-MFMA part of the loop costs under the power governor.
set -e
cd "$(dirname "$0")/../landiff_amd/csrc"
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -fno-slp-vectorize -DLD_ATTN_ABLATE=$m -c ld_attn_p16.hip -o /tmp/ld_attn_p16_ab$m.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblandiff_hip_ablate$m.so $(ls obj/*.o | grep -v ld_attn_p16.o) /tmp/ld_attn_p16_ab$m.o
  echo "built liblandiff_hip_ablate$m.so"
done
