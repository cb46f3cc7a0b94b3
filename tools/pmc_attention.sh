#!/bin/bash
# SQ counter passes of the vision attention launch for both kernels: gpurun_out/<tag>_attn_pmc_v{0,1}_{a,b}
set -e
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2}; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_attn_pmc_v${v}_a -- python3 $R/tools/prof_attention.py $v > $OUT/${TAG}_attn_pmc_v${v}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_attn_pmc_v${v}_b -- python3 $R/tools/prof_attention.py $v > $OUT/${TAG}_attn_pmc_v${v}_b.log 2>&1
done
python3 $R/tools/pmc_summary.py "$OUT/${TAG}_attn_pmc_v*/**/*counter_collection.csv" > $OUT/${TAG}_attn_pmc_summary.txt 2>&1 || true
cat $OUT/${TAG}_attn_pmc_summary.txt
