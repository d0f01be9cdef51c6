#!/bin/bash
# Round 6, visit d: the operand split with non-temporal loads of G (same-box A/B), the fabric traffic of the Gram tile kernel by
# span (PMC FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only), the host path's view test.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_SPLIT_NT=0 BYZ_GRAM_SPLIT_NT=1 > $O/split_nt_ab_n4000.txt 2>&1
cat $O/split_nt_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_SPLIT_NT=0 BYZ_GRAM_SPLIT_NT=1 > $O/split_nt_ab_n10000.txt 2>&1
cat $O/split_nt_ab_n10000.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "view_like or assertions_like" 2>&1 | tail -3
for span in 1 4; do
  export ITERS=2 BYZ_GRAM_KSPAN=$span
  SETS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" bash scripts/gpu_pmc.sh r06d_gram_n4000_span$span gram 4000 1000448 > /dev/null 2>&1
  echo "== span $span"; grep -A12 "gram_planes_kernel" gpurun_out/r06d_gram_n4000_span$span/summary.txt | head -40
done
