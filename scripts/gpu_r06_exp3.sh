#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06bc
mkdir -p $OUT
export TMPDIR=/tmp
export BYZ_BENCH_ONE_DEVICE=1
echo "== rehearsal: self-spawned, 2 ranks on GPU 0 over gloo, small c4"
timeout 600 python bench.py --gpus 2 --clients 1000 --params 400000 --steps 3 --warmup 1 --detail-file $OUT/rehearsal2_detail.json > $OUT/rehearsal2_stdout.txt 2> $OUT/rehearsal2_stderr.txt
echo "rc=$? lines=$(wc -l < $OUT/rehearsal2_stdout.txt)"; tail -n 1 $OUT/rehearsal2_stdout.txt | cut -c1-1500; echo; tail -5 $OUT/rehearsal2_stderr.txt
echo "== rehearsal: the driver's own form, 4 ranks, N = 4000 x 2e6"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --params 2000000 --steps 3 --warmup 1 --detail-file $OUT/rehearsal4_detail.json > $OUT/rehearsal4_stdout.txt 2> $OUT/rehearsal4_stderr.txt
echo "rc=$? lines=$(wc -l < $OUT/rehearsal4_stdout.txt)"; tail -n 1 $OUT/rehearsal4_stdout.txt | cut -c1-1500; echo; tail -5 $OUT/rehearsal4_stderr.txt
echo "== rehearsal: c5u slice form (weak scaling), 2 ranks, reduced columns"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --workload c5u --clients 3000 --params 600000 --steps 2 --warmup 1 --detail-file $OUT/rehearsal_c5u_detail.json > $OUT/rehearsal_c5u_stdout.txt 2> $OUT/rehearsal_c5u_stderr.txt
echo "rc=$? lines=$(wc -l < $OUT/rehearsal_c5u_stdout.txt)"; tail -n 1 $OUT/rehearsal_c5u_stdout.txt | cut -c1-1200; echo; tail -5 $OUT/rehearsal_c5u_stderr.txt
