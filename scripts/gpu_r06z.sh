#!/bin/bash
# Round 6, visit z: the bands inside column panels (BYZ_GRAM_ORDER=2: panels of 32 slabs; >= 8: panels of that many) at N = 10,000,
# where a chunk of planes (335 MB) does not fit the Infinity Cache
set -u
export TMPDIR=/tmp
O=gpurun_out/r06z
mkdir -p $O
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_ORDER=1 BYZ_GRAM_ORDER=2 BYZ_GRAM_ORDER=16 BYZ_GRAM_ORDER=24 BYZ_GRAM_ORDER=40 2>&1 | grep rep > $O/panel_ab_n10000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 7601 401408 BYZ_GRAM_ORDER=1 BYZ_GRAM_ORDER=2 2>&1 | grep rep > $O/panel_ab_n7601.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_ORDER=1 BYZ_GRAM_ORDER=16 2>&1 | grep rep > $O/panel_ab_n4000.txt
cat $O/panel_ab_n10000.txt $O/panel_ab_n7601.txt $O/panel_ab_n4000.txt
