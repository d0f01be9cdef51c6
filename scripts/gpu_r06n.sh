#!/bin/bash
# Round 6, visit n: the 16x16x32 Gram kernel with the instructions behind the barrier in a pinned interleave (BYZ_GRAM_PIN=1)
# against the compiler's order (0), and against the 32x32x16 kernel; same box.  Then the sharded test whose near-tie rule moved.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06n
mkdir -p $O
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_PIN=0 BYZ_GRAM_PIN=1 BYZ_GRAM_MFMA=32 > $O/pin_ab_n4000.txt 2>&1; cat $O/pin_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_PIN=0 BYZ_GRAM_PIN=1 > $O/pin_ab_n10000.txt 2>&1; cat $O/pin_ab_n10000.txt
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_round4.py -m gpu -q -k "columns_layout or skipped_blocks" 2>&1 | tail -3
