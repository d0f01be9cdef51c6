#!/bin/bash
# quick GPU visit: parity tests (optionally filtered) + quick_perf selections
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
K=${K:-}
timeout 900 python -m pytest tests -m gpu -x -q ${K:+-k "$K"} 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 600 python scripts/quick_perf.py "$@" 2>&1 | tee $OUT/perf.txt
