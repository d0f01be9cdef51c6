#!/bin/bash
# round 6, session 3, visit 1: K1 grid cap A/B for the small Krum path; tall-lane trimmed-mean shapes (tests + A/B)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06ba
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tall-lane shapes: trimmed-mean / Bulyan tests under BYZ_TM_LEAN_TALL_LANES=1"
BYZ_TM_LEAN_TALL_LANES=1 timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "trimmed or ring or bulyan or window or lean" 2>&1 | tail -8 | tee $OUT/pytest_tall_lanes.txt
echo "== tall-lane A/B"
timeout 600 python scripts/tm_ab.py BYZ_TM_LEAN_TALL_LANES 0,1 2>&1 | tee $OUT/tm_tall_lanes_ab.txt
echo "== phases (stamps) of the committed shapes and the tall-lane ones"
for v in 0 1; do
  BYZ_TM_LEAN_TALL_LANES=$v BYZ_TM_LEAN_TIMING=1 timeout 300 python scripts/tm_ab.py BYZ_NOTHING 0 2>&1 | grep "^lean" | sort | uniq -c | sort -rn | head -12 | tee -a $OUT/tm_phases_$v.txt
done
echo "== c2 grid cap"
timeout 1500 bash scripts/c2_grid_ab.sh r06ba/c2_grid_ab
