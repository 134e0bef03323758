#!/bin/bash
# Builds liblandiff_hip.so for gfx950 (cross-compiles without a GPU): the shipped library, i.e. the kernels that run.
# LD_BUILD_VARIANTS=1 also builds ../variants/liblandiff_hip_variants.so = the same sources with -DLD_VARIANTS plus the files that
# hold only measured alternatives (128-row attention tile, round-1 attention pipeline, persistent / chained decode step,
# software-pipelined and register-staged GEMM loops): bit-identical, measured slower, kept under test (tests/variants/) and
# loaded by nothing else (LANDIFF_HIP_LIB selects it for a tools/ A/B run).
set -e
cd "$(dirname "$0")"
VARIANT_ONLY="ld_attn_pipe.hip ld_attn_q128.hip ld_llm_fused.hip"
HDRS="ld_common.h ld_attn.h ld_attn_q64_body.h ld_llm_dev.h ../../include/landiff_hip.h build.sh"      # (build.sh: a change of flags rebuilds everything)

build_lib() {   # $1 = object dir, $2 = output, $3 = extra flags, $4.. = sources
  local objdir=$1 out=$2 flags=$3; shift 3
  mkdir -p "$objdir" "$(dirname "$out")"
  local pids=() objs=()
  for s in "$@"; do
    local o=$objdir/${s%.hip}.o
    objs+=("$o")
    local stale=0
    [ ! -f "$o" ] || [ "$s" -nt "$o" ] && stale=1
    for h in $HDRS; do [ "$h" -nt "$o" ] && stale=1; done
    if [ $stale = 1 ]; then
      local extra=""
      # -fno-slp-vectorize: (attention files) the vectoriser's packed fp32 ops are an anti-lever beside MFMAs; (ld_norm.hip, ld_llm*.hip,
      # round 5) it emits v_pk_*_f32 whose LOW lane reads the HIGH register of an operand pair (op_sel) for RoPE-like expressions, and
      # that operand form reads 0.0 in lanes 48-63 when MFMA waves of another kernel share the SIMD (tools/probe/pk_f32_coresidency.hip).
      # tools/audit_pk_f32.py checks the BUILT library for the form (a CPU test), whatever the flags of a file are.
      case $s in ld_attn_pipe.hip|ld_attn_p16.hip|ld_attn_q64.hip|ld_attn_q64_exact.hip|ld_attn_q128.hip|ld_norm.hip) extra="-fno-slp-vectorize";; esac
      # the forms of the decode step (one launch per operation / chained / one persistent launch) must produce the same bits: no
      # implicit mul+add fusion, whose outcome depends on the code around an expression (explicit fmaf / dot2 are unaffected).
      # -fno-slp-vectorize (round 5): the vectoriser turns e.g. RoPE's `a*c - b*s` / `a*s + b*c` into v_pk_mul_f32 / v_pk_add_f32 with
      # op_sel / neg modifiers, and THOSE results were observed to depend, at the level of one fp32 rounding, on what the other waves
      # of the SIMD do: with the DiT's 64-row attention kernel co-resident (AR decode under the DiT loop: generate_many, the streaming
      # loop) the prefill's rotated q / k came out one bf16 step off in ~1e-5 .. 1e-3 of the even elements, and the decode sampled
      # other tokens (tools/llm_race_probe2.py, profiles/r05_llm_packed_f32_under_coresidency.txt).  Scalar v_mul / v_sub / v_add: clean.
      case $s in ld_llm.hip|ld_llm_fused.hip) extra="-ffp-contract=off -fno-slp-vectorize";; esac
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result $extra $flags -c "$s" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait $p; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
  echo "built $out"
}

LEAN=""
for s in $(ls ld_*.hip); do
  case " $VARIANT_ONLY " in *" $s "*) ;; *) LEAN="$LEAN $s";; esac
done
build_lib obj ../liblandiff_hip.so "" $LEAN
if [ "${LD_BUILD_VARIANTS:-0}" = "1" ]; then
  build_lib obj_variants ../variants/liblandiff_hip_variants.so "-DLD_VARIANTS=1" $(ls ld_*.hip)
fi
