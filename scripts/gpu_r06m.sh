#!/bin/bash
# Round 6, visit m: the whole GPU suite on the 16x16x32 Gram kernel, then its counters (separate PMC passes, kernel trace only).
set -u
export TMPDIR=/tmp
O=gpurun_out/r06m
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
for shape in 16 32; do
  export ITERS=2 BYZ_GRAM_MFMA=$shape
  SETS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" bash scripts/gpu_pmc.sh r06m_gram_n4000_mfma$shape gram 4000 1000448 > /dev/null 2>&1
  echo "== MFMA shape $shape"; grep -A12 "gram_planes" gpurun_out/r06m_gram_n4000_mfma$shape/summary.txt | grep -v "reduce\|row_sig\|candidate\|verify\|compact" | head -40
done
