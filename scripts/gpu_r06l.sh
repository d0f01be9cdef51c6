#!/bin/bash
# Round 6, visit l: the f16x2 Gram tile kernel on v_mfma_f32_16x16x32_f16 (default) against the 32x32x16 form (BYZ_GRAM_MFMA=32):
# the Gram / plane / distance tests on the new kernel, then the same-box A/B at one launch of configs[3] and configs[4]'s slice.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06l
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py tests/test_large_golden.py tests/test_gpu_sharded.py -m gpu -x -q -k "gram or plane or f16x2 or sampled or long_k or large or fullsize or twin or outlier or share or config4 or distances" > $O/pytest_gram16.txt 2>&1; tail -5 $O/pytest_gram16.txt
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_MFMA=32 BYZ_GRAM_MFMA=16 > $O/mfma_shape_ab_n4000.txt 2>&1; cat $O/mfma_shape_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_MFMA=32 BYZ_GRAM_MFMA=16 > $O/mfma_shape_ab_n10000.txt 2>&1; cat $O/mfma_shape_ab_n10000.txt
