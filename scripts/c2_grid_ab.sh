#!/bin/bash
# Same-box A/B of K1's grid cap (BYZ_KRUM_SMALL_GRID) for the N <= 128 Krum round: wall per round over 2000 raw C-ABI rounds,
# three alternations.  Usage (GPU box, repo root): bash scripts/c2_grid_ab.sh <tag>
TAG=${1:-c2grid}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG.txt
: > $OUT
python scripts/c2_rounds.py 79510 500 > /dev/null 2>&1   # (the box's first import and clocks)
for rep in 1 2 3; do
  for d in 79510 21840 117706; do
    for cap in 256 208 160 128 104 80 64; do
      echo -n "rep $rep cap $cap " >> $OUT
      BYZ_KRUM_SMALL_GRID=$cap python scripts/c2_rounds.py $d 2000 2>&1 | tail -1 >> $OUT
    done
  done
done
for d in 79510 21840; do
  for cap in 256 128; do
    for skip in 1 2; do
      echo -n "skip $skip cap $cap " >> $OUT
      BYZ_KRUM_SMALL_SKIP=$skip BYZ_KRUM_SMALL_GRID=$cap python scripts/c2_rounds.py $d 2000 2>&1 | tail -1 >> $OUT
    done
  done
done
cat $OUT
