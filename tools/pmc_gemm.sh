#!/bin/bash
# rocprofv3 --pmc passes (SQ / LDS / FETCH_SIZE, one pass each) over the four DiT GEMM shapes for LD_GEMM_8P=0 and 1 -> $1
export TMPDIR=/tmp
out=$1; mkdir -p gpurun_out/pmc
: > $out
for v in 0 1; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    d=gpurun_out/pmc/p_$v_$(echo $set | cut -c1-12 | tr ' ' _)
    rm -rf $d
    LD_GEMM_8P=$v rocprofv3 --pmc $set -f csv -d $d -- python3 tools/gemm_shapes_only.py > $d.log 2>&1
    echo "=== LD_GEMM_8P=$v : $set" >> $out
    python3 tools/pmc_parse.py $d >> $out 2>&1
  done
done
rm -rf gpurun_out/pmc
