#!/bin/bash
OUT=gpurun_out/r02i; mkdir -p $OUT
timeout 300 python -u scripts/r02_perf.py loop 2>&1 | tail -8 | tee $OUT/perf.txt
timeout 300 python -u -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 150 -k "bulyan or selection or config" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt | cut -c1-250
