#!/bin/bash
OUT=gpurun_out/r02l; mkdir -p $OUT
timeout 200 python -u scripts/r02_perf.py tm 2>&1 | tail -16 | tee $OUT/perf.txt
timeout 400 python -u -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q --timeout 150 -k "trimmed or ring or median or window or tie or golden or bulyan or config" > $OUT/pytest.txt 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt | cut -c1-250 | tail -30
