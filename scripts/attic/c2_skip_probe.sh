#!/bin/bash
# What do the two launches of a small-Krum round cost?  BYZ_KRUM_SMALL_SKIP drops kernels (timing only): bit 0 K1, bit 1 K2.
cd ${GRAFT_REPO_ROOT:-.}
for d in 79510 21840; do
  for skip in 0 2 1; do
    echo -n "skip=$skip "; BYZ_KRUM_SMALL_SKIP=$skip timeout 120 python scripts/c2_rounds.py $d 2000 2>&1 | tail -1
  done
done
