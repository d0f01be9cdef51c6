#!/bin/bash
# Round 5, visit a: whole GPU suite, the bench exactly as the driver runs it (wall-clocked, line size checked), the Gram
# block-skip A/B on one box, the sequential attack statistics at configs[4]'s slice.
set -u
TAG=${1:-r05a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 --durations=12 2>&1 | tail -70 > $OUT/pytest_gpu.txt
tail -40 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
START=$(date +%s.%N)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
echo "bench rc=$? wall=$(echo "$(date +%s.%N) - $START" | bc 2>/dev/null) s; stdout $(wc -c < $OUT/bench_stdout.txt) B in $(wc -l < $OUT/bench_stdout.txt) line(s); stderr $(wc -c < $OUT/bench_stderr.txt) B" | tee $OUT/bench_wall.txt
tail -c 4200 $OUT/bench_stdout.txt
tail -5 $OUT/bench_stderr.txt
timeout 600 python scripts/gram_ab.py BYZ_GRAM_BLOCK_SKIP=0,BYZ_GRAM_BLOCK_SKIP=1 4000 262224 2>&1 | tail -6 | tee $OUT/gram_skip_ab_n4000.txt
timeout 600 python scripts/gram_ab.py BYZ_GRAM_BLOCK_SKIP=0,BYZ_GRAM_BLOCK_SKIP=1 10000 98384 2>&1 | tail -6 | tee $OUT/gram_skip_ab_n10000.txt
timeout 300 python bench.py --workload attack --clients 2400 --params 3125000 --steps 5 --warmup 2 --no-cpu-baseline --detail-file $OUT/attack_detail.json 2>/dev/null | tee $OUT/attack_line.json
