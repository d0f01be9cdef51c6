#!/bin/bash
# Round 6, visit j: the wall time of a configs[2] trimmed-mean round and of a configs[1] Krum round with the dynamic-LDS
# attribute set once per (kernel, context) (libbyzagg.so) against once per launch (libbyzagg_prev.so), processes alternated.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06j
mkdir -p $O
for rep in 1 2; do
  for lib in libbyzagg_prev.so libbyzagg.so; do
    for steps in 10 200; do
      echo "== $lib c3 steps=$steps (pass $rep)" >> $O/attr_ab.txt
      BYZ_LIBRARY=$PWD/attacking_federate_learning_amd/$lib timeout 300 python bench.py --workload c3 --no-cpu-baseline --steps $steps --warmup 5 --detail-file '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernels', d.get('kernels_ms'), 'frac', d['roofline']['frac'])" >> $O/attr_ab.txt 2>&1
    done
    echo "== $lib c4-small (N=4000, D=262144) steps=5 (pass $rep)" >> $O/attr_ab.txt
    BYZ_LIBRARY=$PWD/attacking_federate_learning_amd/$lib timeout 300 python bench.py --clients 4000 --params 262144 --no-cpu-baseline --no-extras --no-sharded-w1 --steps 5 --warmup 2 --detail-file '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernels', d.get('kernels_ms'))" >> $O/attr_ab.txt 2>&1
  done
done
cat $O/attr_ab.txt
