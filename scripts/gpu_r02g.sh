#!/bin/bash
OUT=gpurun_out/r02g; mkdir -p $OUT
timeout 200 python -u scripts/r02_perf.py gram 2>&1 | tail -12 | tee $OUT/perf.txt
timeout 250 python -u -m pytest tests/test_gpu_scale.py -m gpu -v -s --timeout 100 -k "plane_gram or f16x2" > $OUT/pytest.txt 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed|assert|Error" $OUT/pytest.txt | cut -c1-250 | tail -20
