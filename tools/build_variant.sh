#!/bin/bash
# builds landiff_amd/variants/lib_<name>.so = the library with extra -D flags on ld_gemm.hip (compile-time experiments; tools only)
# usage: tools/build_variant.sh <name> <file.hip> -DFLAG ...
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../landiff_amd/csrc"
mkdir -p ../variants /tmp/ldvar_$name
extra=""
case $file in ld_attn_pipe.hip|ld_attn_p16.hip|ld_attn_q64.hip|ld_attn_q64_exact.hip|ld_attn_q128.hip|ld_norm.hip) extra="-fno-slp-vectorize";; ld_llm.hip|ld_llm_fused.hip) extra="-ffp-contract=off -fno-slp-vectorize";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result $extra "$@" -c $file -o /tmp/ldvar_$name/${file%.hip}.o
objs=""
for o in obj/*.o; do b=$(basename $o); if [ "$b" = "${file%.hip}.o" ]; then objs="$objs /tmp/ldvar_$name/$b"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$name.so $objs
echo built landiff_amd/variants/lib_$name.so
