#!/bin/bash
OUT=gpurun_out/r02e; mkdir -p $OUT
timeout 200 python scripts/r02_perf.py gram variants 2>&1 | tail -30 | tee $OUT/perf.txt
BYZ_GRAM_PLANES_VARIANT=3 timeout 150 python -u -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 60 -x -k "plane_gram" > $OUT/plane_test.txt 2>&1
tail -5 $OUT/plane_test.txt | cut -c1-300
