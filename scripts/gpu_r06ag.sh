#!/bin/bash
# Round 6, visit ag: the speculative Bulyan loop (batches of picks decided optimistically, their contested ones verified together)
# against the loop of rounds 2-5 (BYZ_BULYAN_BATCH=0): same box, selections compared; then the selection tests on the new default.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06ag
mkdir -p $O
export BYZ_BULYAN_STATS=1
REPS=2 timeout 600 python scripts/bulyan_loop_ab.py 4000 BYZ_BULYAN_BATCH=0 BYZ_BULYAN_BATCH=16 BYZ_BULYAN_BATCH=8 BYZ_BULYAN_BATCH=4 BYZ_BULYAN_BATCH=1 2>&1 | grep -v amdgpu.ids | tee $O/spec_ab_n4000.txt
REPS=2 timeout 900 python scripts/bulyan_loop_ab.py 10000 BYZ_BULYAN_BATCH=0 BYZ_BULYAN_BATCH=16 BYZ_BULYAN_BATCH=8 2>&1 | grep -v amdgpu.ids | tee $O/spec_ab_n10000.txt
ATTACK=1 REPS=2 timeout 900 python scripts/bulyan_loop_ab.py 10000 BYZ_BULYAN_BATCH=0 BYZ_BULYAN_BATCH=16 2>&1 | grep -v amdgpu.ids | tee $O/spec_ab_n10000_attack.txt
ATTACK=1 REPS=1 timeout 900 python scripts/bulyan_loop_ab.py 4000 BYZ_BULYAN_BATCH=0 BYZ_BULYAN_BATCH=16 2>&1 | grep -v amdgpu.ids | tee $O/spec_ab_n4000_attack.txt
REPS=1 timeout 900 python scripts/bulyan_loop_ab.py 700 BYZ_BULYAN_BATCH=0 BYZ_BULYAN_BATCH=16 BYZ_BULYAN_BATCH=3 2>&1 | grep -v amdgpu.ids | tee $O/spec_ab_n700.txt
unset BYZ_BULYAN_STATS
timeout 1500 python -m pytest tests -m gpu -q -x -k "bulyan or selection or select or krum or golden or large or scale or parity" 2>&1 | tail -5
