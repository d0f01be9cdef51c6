#!/bin/bash
# Round 6, visit w: on the band order -- runs of units claimed from a counter (BYZ_GRAM_CLAIM=1) against dealt in turn; twice the
# plane budget (half the launches); the bitwise test of orders / claims.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06w
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "skipped_blocks" 2>&1 | tail -3
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_CLAIM=0 BYZ_GRAM_CLAIM=1 BYZ_GRAM_ORDER=0 2>&1 | grep rep > $O/claim_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_CLAIM=0 BYZ_GRAM_CLAIM=1 2>&1 | grep rep > $O/claim_ab_n10000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 4000 2000896 BYZ_GRAM_PLANE_MB=16384 BYZ_GRAM_PLANE_MB=32768 BYZ_GRAM_PLANE_MB=32768,BYZ_GRAM_CLAIM=1 2>&1 | grep rep > $O/budget_ab_n4000.txt
cat $O/claim_ab_n4000.txt $O/claim_ab_n10000.txt $O/budget_ab_n4000.txt
