#!/bin/bash
# Round 6, visit c: same-box A/B of the trimmed mean's sweep B -- predicates as lane masks (libbyzagg.so) against round 5's
# booleans (libbyzagg_tm_r05.so: the same sources with -DBYZ_TM_SWEEP_B_BOOLEANS) -- two processes alternated on one box,
# (libbyzagg_tm_r05.so = window_lean.hip of commit a36073c compiled with -DBYZ_TM_SWEEP_B_BOOLEANS and linked with the other objects; the macro
# left the source again once this A/B was recorded: EXPERIMENTS.md T3)
# then the trimmed-mean GPU tests on the new library.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
for rep in 1 2 3; do
  for lib in libbyzagg_tm_r05.so libbyzagg.so; do
    echo "== $lib (pass $rep)" >> $O/tm_sweep_b_ab.txt
    BYZ_LIBRARY=$PWD/attacking_federate_learning_amd/$lib timeout 300 python scripts/tm_ab.py BYZ_TM_LEAN_XCD 1 >> $O/tm_sweep_b_ab.txt 2>&1
  done
done
cat $O/tm_sweep_b_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "trimmed or ring or bulyan or tm or golden" > $O/pytest_tm.txt 2>&1
tail -3 $O/pytest_tm.txt
