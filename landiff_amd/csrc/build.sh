#!/bin/bash
# Builds liblandiff_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../liblandiff_hip.so
SRCS=$(ls ld_*.hip)
mkdir -p obj
pids=()
for s in $SRCS; do
  o=obj/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ ld_common.h -nt "$o" ] || [ ld_attn.h -nt "$o" ] || [ ld_llm_dev.h -nt "$o" ] || [ ../../include/landiff_hip.h -nt "$o" ]; then
    extra=""
    { [ "$s" = "ld_attn_pipe.hip" ] || [ "$s" = "ld_attn_p16.hip" ] || [ "$s" = "ld_attn_q64.hip" ] || [ "$s" = "ld_attn_q128.hip" ]; } && extra="-fno-slp-vectorize"
    # the two forms of the decode step (one launch per operation / one persistent launch) must produce the same bits: no
    # implicit mul+add fusion, whose outcome depends on the code around an expression (explicit fmaf / dot2 are unaffected)
    { [ "$s" = "ld_llm.hip" ] || [ "$s" = "ld_llm_fused.hip" ]; } && extra="-ffp-contract=off"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result $extra -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT obj/*.o
echo "built $OUT"
