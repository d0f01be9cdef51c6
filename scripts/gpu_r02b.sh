#!/bin/bash
# r02b: parity suite + scale tests on the new selection / distance kernels
OUT=gpurun_out/r02b; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 100 2>&1 | tail -60 > $OUT/parity.txt
tail -3 $OUT/parity.txt
timeout 500 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 160 --durations=20 -s 2>&1 | tail -150 > $OUT/scale.txt
tail -40 $OUT/scale.txt
