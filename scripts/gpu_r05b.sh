#!/bin/bash
# Round 5, visit b: the register-resident attack statistics -- parity first (under a short timeout: a chain that never gets
# its turn must not cost the box), then its time against the two-pass kernel on one box.
set -u
TAG=${1:-r05b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pipeline_golden.py tests/test_baseline_golden.py tests/test_gpu_round4.py -m gpu -q --timeout 120 -x 2>&1 | tail -25 | tee $OUT/pytest_attack.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q --timeout 120 -k "attack or drift or config5 or device_server or golden" 2>&1 | tail -8 | tee -a $OUT/pytest_attack.txt
for spec in "2400 3125000" "2400 1000000" "240 8000000" "640 4000000" "1000 2000000"; do
  set -- $spec
  for mode in 1 0; do
    echo "m=$1 D=$2 BYZ_ATTACK_RESIDENT=$mode"
    BYZ_ATTACK_RESIDENT=$mode timeout 300 python bench.py --workload attack --clients $1 --params $2 --steps 10 --warmup 2 --no-cpu-baseline --detail-file $OUT/attack_${1}_${2}_$mode.json 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('   %.3f ms per round, frac of HBM %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
  done
done 2>&1 | tee $OUT/attack_ab.txt
