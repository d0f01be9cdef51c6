#!/bin/bash
# Round 6, visit a: the new GPU tests (Gram spans bitwise, the attack's redo path), then the span A/B of the deferred Gram tile
# kernel on one box at one launch of configs[3] and of configs[4]'s slice.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_pipeline_golden.py -m gpu -x -q -k "skipped_blocks or redo_path or attack_statistics_bit" > $O/pytest_new.txt 2>&1
tail -3 $O/pytest_new.txt
REPS=3 CALLS=3 timeout 900 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_KSPAN=1 BYZ_GRAM_KSPAN=2 BYZ_GRAM_KSPAN=4 BYZ_GRAM_KSPAN=8 BYZ_GRAM_KSPAN=16 BYZ_GRAM_KSPAN=4,BYZ_GRAM_ROUND=0 BYZ_GRAM_KSPAN=1,BYZ_GRAM_ROUND=0 > $O/span_ab_n4000.txt 2>&1
cat $O/span_ab_n4000.txt
REPS=2 CALLS=2 timeout 900 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_KSPAN=1 BYZ_GRAM_KSPAN=2 BYZ_GRAM_KSPAN=4 BYZ_GRAM_KSPAN=7 BYZ_GRAM_KSPAN=4,BYZ_GRAM_ROUND=0 > $O/span_ab_n10000.txt 2>&1
cat $O/span_ab_n10000.txt
