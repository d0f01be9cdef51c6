#!/bin/bash
# Round 6, visit s: the pinned steady state in trips of eight super-stages (C = 0 at the head of a chain, the flush inside the
# pinned groups, the first half pinned: libbyzagg.so) against the two-super-stage pinned form (libbyzagg_prev.so), processes
# alternated on one box; bitwise against the compiler's order; the Gram tests.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06s
mkdir -p $O
for rep in 1 2 3; do
  for lib in libbyzagg_prev.so libbyzagg.so; do
    echo "== $lib (pass $rep)" >> $O/trips_ab.txt
    BYZ_LIBRARY=$PWD/attacking_federate_learning_amd/$lib REPS=2 CALLS=3 timeout 300 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_PIN=0 BYZ_GRAM_PIN=1 2>&1 | grep rep >> $O/trips_ab.txt
  done
done
BYZ_LIBRARY=$PWD/attacking_federate_learning_amd/libbyzagg_prev.so REPS=2 CALLS=2 timeout 300 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_PIN=1 2>&1 | grep rep >> $O/trips_ab.txt
REPS=2 CALLS=2 timeout 300 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_PIN=1 2>&1 | grep rep >> $O/trips_ab.txt
cat $O/trips_ab.txt
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py tests/test_large_golden.py tests/test_gpu_sharded.py -m gpu -q -k "gram or plane or f16x2 or sampled or long_k or large or fullsize or twin or outlier or share or config4 or distances or columns_layout" 2>&1 | tail -3
