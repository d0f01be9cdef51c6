#!/bin/bash
# Round 6, visit f: the Bulyan loop with and without the re-score's skip of the all-marks batches at the front of a row
# (BYZ_BULYAN_FRONT=0 / 1), same box, alternated; then every selection test.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
REPS=3 timeout 600 python scripts/bulyan_loop_ab.py 4000 BYZ_BULYAN_FRONT=0 BYZ_BULYAN_FRONT=1 > $O/loop_front_n4000.txt 2>&1; cat $O/loop_front_n4000.txt
REPS=2 timeout 900 python scripts/bulyan_loop_ab.py 10000 BYZ_BULYAN_FRONT=0 BYZ_BULYAN_FRONT=1 > $O/loop_front_n10000.txt 2>&1; cat $O/loop_front_n10000.txt
ATTACK=1 REPS=2 timeout 900 python scripts/bulyan_loop_ab.py 10000 BYZ_BULYAN_FRONT=0 BYZ_BULYAN_FRONT=1 > $O/loop_front_n10000_attack.txt 2>&1; cat $O/loop_front_n10000_attack.txt
timeout 1200 python -m pytest tests -m gpu -q -k "bulyan or selection or config4 or config5 or large or golden or rescore" > $O/pytest_selection.txt 2>&1; tail -3 $O/pytest_selection.txt
