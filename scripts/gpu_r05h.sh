#!/bin/bash
# Round 5, visit h: rocprofv3 kernel stats of the two north-star legs (one GPU's slice of configs[4]) as standalone workloads
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $OUT
for w in c5s c5u; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o $w --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --detail-file $OUT/${w}_detail.json > $OUT/$w.log 2>&1 )
  tail -1 $OUT/$w.log > $OUT/${w}_line.json
  for f in $(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${w}_kernel_stats.csv; head -12 $f | cut -c1-160; done
done
find $OUT -name '*kernel_trace.csv' -delete
