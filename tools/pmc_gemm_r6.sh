#!/bin/bash
# rocprofv3 --pmc passes (SQ, LDS, GRBM: one pass each) over the four DiT GEMM shapes for two builds of the library -- the round-6
# two-phase loop (shipped) and -DLD_GEMM_PH4 (rounds 3-5: four phases; tools/build_variant.sh ph4 ld_gemm.hip -DLD_GEMM_PH4=1) -> $1
export TMPDIR=/tmp
out=$1; mkdir -p gpurun_out/pmc
: > $out
for v in ph4 shipped; do
  lib=""; [ $v = ph4 ] && lib=$PWD/landiff_amd/variants/lib_ph4.so
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
             "GRBM_GUI_ACTIVE"; do
    d=gpurun_out/pmc/p_${v}_$(echo $set | cut -c1-12 | tr ' ' _)
    rm -rf $d
    if [ -n "$lib" ]; then LANDIFF_HIP_LIB=$lib rocprofv3 --pmc $set -f csv -d $d -- python3 tools/gemm_shapes_only.py > $d.log 2>&1
    else rocprofv3 --pmc $set -f csv -d $d -- python3 tools/gemm_shapes_only.py > $d.log 2>&1; fi
    echo "=== $v : $set" >> $out
    python3 tools/pmc_parse.py $d | grep -A8 "ld_gemm8p_kernel" >> $out 2>&1
  done
done
rm -rf gpurun_out/pmc
