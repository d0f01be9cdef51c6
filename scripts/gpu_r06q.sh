#!/bin/bash
# Round 6, visit q: the 16x16x32 Gram kernel with 512-column MFMA chains (BYZ_GRAM_FLUSH16=32, the default) against 256-column
# ones (16): speed on one box, and the accuracy tests under both.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06q
mkdir -p $O
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_FLUSH16=16 BYZ_GRAM_FLUSH16=32 > $O/flush_ab_n4000.txt 2>&1; cat $O/flush_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_FLUSH16=16 BYZ_GRAM_FLUSH16=32 > $O/flush_ab_n10000.txt 2>&1; cat $O/flush_ab_n10000.txt
for fl in 16 32; do
  echo "== accuracy tests with BYZ_GRAM_FLUSH16=$fl"
  BYZ_GRAM_FLUSH16=$fl timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py tests/test_large_golden.py -m gpu -q -s -k "f16x2 or skipped_blocks or fullsize or large or sampled" 2>&1 | grep -i "passed\|failed\|max\|worst\|error" | tail -12
done
