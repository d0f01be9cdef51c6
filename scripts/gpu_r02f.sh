#!/bin/bash
OUT=gpurun_out/r02f; mkdir -p $OUT
for v in 0 3; do
for k in "2900-24676-0" "3000-40960-0" "3300-17444-700" "4000-163876-0"; do
  s=$(date +%s.%N)
  BYZ_GRAM_PLANES_VARIANT=$v timeout 60 python -u -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 50 -x -k "plane_gram and $k" > $OUT/t_$v_$k.txt 2>&1
  rc=$?
  e=$(date +%s.%N)
  echo "variant $v case $k rc=$rc elapsed $(echo "$e - $s" | bc) : $(tail -1 $OUT/t_$v_$k.txt | cut -c1-150)" | tee -a $OUT/summary.txt
done
done
