#!/bin/bash
# Round 5, visit d: the deferred slab update of the long-K Gram -- bitwise tests, then the same-box A/B, then the c4 bench line
set -u
TAG=${1:-r05d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_scale.py tests/test_gpu_sharded.py tests/test_pipeline_golden.py -m gpu -q --timeout 300 -x 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 600 python scripts/gram_ab.py BYZ_GRAM_DEFER=0,BYZ_GRAM_DEFER=1 4000 262224 2>&1 | tail -5 | tee $OUT/gram_defer_ab_n4000.txt
timeout 600 python scripts/gram_ab.py BYZ_GRAM_DEFER=0,BYZ_GRAM_DEFER=1 10000 98384 2>&1 | tail -5 | tee $OUT/gram_defer_ab_n10000.txt
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-sharded-w1 --detail-file $OUT/bench_detail.json 2>/dev/null | tail -1 | tee $OUT/bench_line.json | cut -c1-1500
