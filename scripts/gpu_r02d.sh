#!/bin/bash
# r02d: plane Gram parity + timings, Bulyan loop timings by band
OUT=gpurun_out/r02d; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 200 -k "plane_gram" 2>&1 | tail -30 > $OUT/plane_test.txt
tail -30 $OUT/plane_test.txt
timeout 400 python scripts/r02_perf.py gram loop 2>&1 | tail -30 | tee $OUT/perf.txt
