#!/bin/bash
# What does one dependent launch cost at the end of a small-Krum round?  BYZ_KRUM_SMALL_SKIP drops kernels (timing only).
cd ${GRAFT_REPO_ROOT:-.}
for d in 79510 21840; do
  for skip in 0 16 24 28 30 1 31; do
    echo -n "skip=$skip "; BYZ_KRUM_SMALL_SKIP=$skip timeout 120 python scripts/c2_rounds.py $d 2000 2>&1 | tail -1
  done
done
