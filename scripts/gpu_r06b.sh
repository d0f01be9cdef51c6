#!/bin/bash
# Round 6, visit b: the whole GPU suite with the new tests (large reference goldens, the margin protocol's second clause).
set -u
export TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
rm -f gpurun_out/margin_protocol.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
cat gpurun_out/margin_protocol.jsonl
