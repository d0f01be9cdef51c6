#!/bin/bash
# One full GPU visit: whole GPU suite (no -x: every failure listed), smoke, contract bench, kernel trace of the same command
set -u
TAG=${1:-visit}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 200 --durations=15 2>&1 | tail -80 > $OUT/pytest_gpu.txt
tail -45 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python bench.py 2> $OUT/bench_c4.err | tee $OUT/bench_c4.json
tail -5 $OUT/bench_c4.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c4 -o c4 --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$OUT/prof_c4.log 2>&1 )
tail -3 $OUT/prof_c4.log
for f in $(find $OUT/prof_c4 -name '*kernel_stats.csv' | head -1); do head -25 $f; done
find $OUT/prof_c4 -name '*kernel_trace.csv' -delete
