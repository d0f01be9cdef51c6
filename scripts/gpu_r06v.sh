#!/bin/bash
# Round 6, visit v: the tile list in bands of four 256-row blocks with rounds of 32 consecutive (chunk, tile) units dealt to the
# XCDs in turn (BYZ_GRAM_ORDER=1) against rounds 2-5's super-block list with a contiguous share per XCD (BYZ_GRAM_ORDER=0):
# same box, alternated, bitwise; then the fabric bytes of both by PMC.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06v
mkdir -p $O
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_ORDER=0 BYZ_GRAM_ORDER=1 2>&1 | grep rep > $O/order_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_ORDER=0 BYZ_GRAM_ORDER=1 2>&1 | grep rep > $O/order_ab_n10000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 7601 401408 BYZ_GRAM_ORDER=0 BYZ_GRAM_ORDER=1 2>&1 | grep rep > $O/order_ab_n7601.txt
cat $O/order_ab_n4000.txt $O/order_ab_n10000.txt $O/order_ab_n7601.txt
for order in 0 1; do
  export ITERS=2 BYZ_GRAM_ORDER=$order
  SETS="FETCH_SIZE;SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" bash scripts/gpu_pmc.sh r06v_gram_n4000_order$order gram 4000 1000448 > /dev/null 2>&1
  echo "== order $order, N = 4000"; grep -A9 "gram_planes16" gpurun_out/r06v_gram_n4000_order$order/summary.txt | grep -v "reduce\|row_sig\|candidate\|verify\|compact" | head -24
  SETS="FETCH_SIZE" bash scripts/gpu_pmc.sh r06v_gram_n10000_order$order gram 10000 401408 > /dev/null 2>&1
  echo "== order $order, N = 10,000"; grep -A2 "gram_planes16" gpurun_out/r06v_gram_n10000_order$order/summary.txt | head -4
done
unset BYZ_GRAM_ORDER
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py tests/test_large_golden.py tests/test_gpu_sharded.py -m gpu -q -k "gram or plane or f16x2 or sampled or long_k or large or fullsize or twin or outlier or share or config4 or distances or columns_layout" 2>&1 | tail -3
