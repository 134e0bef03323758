#!/bin/bash
# rocprofv3 --pmc passes (SQ, LDS, GRBM: one pass each) over the four DiT GEMM shapes on MXFP8 operands (ld_gemm8p_mx_kernel) -> $1
export TMPDIR=/tmp
out=$1; mkdir -p gpurun_out/pmc
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  d=gpurun_out/pmc/p_mx_$(echo $set | cut -c1-12 | tr ' ' _)
  rm -rf $d
  rocprofv3 --pmc $set -f csv -d $d -- python3 tools/gemm_mx_shapes_only.py > $d.log 2>&1
  echo "=== mx : $set" >> $out
  python3 tools/pmc_parse.py $d | grep -A8 "ld_gemm8p_mx_kernel" >> $out 2>&1
done
rm -rf gpurun_out/pmc
