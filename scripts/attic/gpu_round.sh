#!/bin/bash
# One GPU visit: parity tests, contract bench, kernel-trace profile of the same command.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/log.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke" | tee -a $OUT/log.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== bench (default: c4)" | tee -a $OUT/log.txt
timeout 900 python bench.py 2> $OUT/bench_c4.err | tee $OUT/bench_c4.json
tail -5 $OUT/bench_c4.err
echo "== rocprofv3 kernel trace of the same command"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c4 -o c4 --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$OUT/prof_c4.log 2>&1 )
tail -3 $OUT/prof_c4.log
find $OUT/prof_c4 -name '*kernel_stats.csv' | head -3
for f in $(find $OUT/prof_c4 -name '*kernel_stats.csv' | head -1); do head -20 $f; done
# the raw kernel trace is large: keep the stats only
find $OUT/prof_c4 -name '*kernel_trace.csv' -delete
