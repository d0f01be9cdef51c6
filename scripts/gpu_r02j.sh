#!/bin/bash
OUT=gpurun_out/r02j; mkdir -p $OUT
timeout 300 python -u scripts/r02_perf.py loop tm 2>&1 | tail -26 | tee $OUT/perf.txt
timeout 500 python -u -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q --timeout 150 -k "bulyan or selection or config or trimmed or ring or median or window or tie or golden" > $OUT/pytest.txt 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt | cut -c1-250 | tail -30
