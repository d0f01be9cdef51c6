#!/bin/bash
# Round 6, visit y: claimed runs with spare rounds per XCD (a faster XCD takes more runs)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06y
mkdir -p $O
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_SPARE=0 BYZ_GRAM_SPARE=5 BYZ_GRAM_SPARE=12 BYZ_GRAM_SPARE=30 BYZ_GRAM_SPARE=100 2>&1 | grep rep > $O/spare_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_SPARE=0 BYZ_GRAM_SPARE=5 BYZ_GRAM_SPARE=12 BYZ_GRAM_SPARE=30 2>&1 | grep rep > $O/spare_ab_n10000.txt
cat $O/spare_ab_n4000.txt $O/spare_ab_n10000.txt
