#!/bin/bash
# Round 6, visit o: pinned orders 1 .. 4 of the 16x16x32 Gram kernel (libraries built with -DBYZ_GRAM_PIN_ORDER=n), processes
# alternated on one box, one launch of configs[3].
set -u
export TMPDIR=/tmp
O=gpurun_out/r06o
mkdir -p $O
for rep in 1 2; do
  for lib in libbyzagg.so libbyzagg_pin2.so libbyzagg_pin3.so libbyzagg_pin4.so; do
    echo "== $lib (pass $rep)" >> $O/pin_orders.txt
    BYZ_LIBRARY=$PWD/attacking_federate_learning_amd/$lib REPS=2 CALLS=3 timeout 300 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_PIN=1 2>&1 | grep rep >> $O/pin_orders.txt
  done
done
cat $O/pin_orders.txt
