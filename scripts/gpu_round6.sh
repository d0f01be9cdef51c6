#!/bin/bash
# One GPU visit of round 6: the whole GPU suite, smoke, the contract bench exactly as the driver runs it (wall-clocked; the
# stdout line's size checked), the rocprofv3 kernel stats of the same workload, the HBM-traffic PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs, kernel-trace only) and the sanitized torch-free subset.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round6.sh <tag> [skip-tests]
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "${2:-}" != "skip-tests" ]; then
  # (--timeout must stay far above what a COLD box needs for its first `import torch` (1-2 minutes): pytest-timeout's alarm in the
  # middle of that import corrupts it and the interpreter dies with a core dump -- seen once with --timeout 120, reproduced with
  # --timeout 4.  The driver's own command has no --timeout.)
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -m gpu -q --timeout 300 --durations=10 2>&1 | tail -40 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
  echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
fi
echo "== bench as the driver runs it"
T0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s; stdout $(wc -c < $OUT/bench_stdout.txt) B in $(wc -l < $OUT/bench_stdout.txt) line(s); stderr $(wc -c < $OUT/bench_stderr.txt) B" | tee $OUT/bench_wall.txt
tail -n 1 $OUT/bench_stdout.txt > $OUT/bench_line.json; cut -c1-600 $OUT/bench_line.json; echo
echo "== rocprofv3 kernel stats of the c4 workload"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -o c4 --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-sharded-w1 --detail-file $OUT/prof_c4_detail.json > $OUT/prof_c4.log 2>&1 )
for f in $(find $OUT/prof_c4 -name '*kernel_stats.csv' | head -1); do cp $f $OUT/c4_kernel_stats.csv; head -14 $f; done
echo "== rocprofv3 kernel stats of the attack (m = 2400, D = 3.125e6) and of a small-Krum round"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_attack -o a --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --workload attack --clients 2400 --params 3125000 --no-cpu-baseline --steps 10 --warmup 2 --detail-file '' > $OUT/prof_attack.log 2>&1 )
for f in $(find $OUT/prof_attack -name '*kernel_stats.csv' | head -1); do cp $f $OUT/attack_kernel_stats.csv; head -4 $f; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o c2 --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --workload c2 --no-cpu-baseline --steps 200 --warmup 20 --detail-file '' > $OUT/prof_c2.log 2>&1 )
for f in $(find $OUT/prof_c2 -name '*kernel_stats.csv' | head -1); do cp $f $OUT/c2_kernel_stats.csv; head -6 $f; done
find $OUT -name '*kernel_trace.csv' -delete
echo "== PMC traffic passes"
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/c4.$ctr -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-sharded-w1 --steps 1 --warmup 0 --detail-file '' > $OUT/c4.$ctr.log 2>&1
  timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/c3.$ctr -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --no-cpu-baseline --steps 3 --warmup 1 --detail-file '' > $OUT/c3.$ctr.log 2>&1
  timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/c2.$ctr -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --no-cpu-baseline --steps 3 --warmup 1 --detail-file '' > $OUT/c2.$ctr.log 2>&1
  timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/attack.$ctr -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload attack --clients 2400 --params 1000000 --no-cpu-baseline --steps 3 --warmup 1 --detail-file '' > $OUT/attack.$ctr.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(out + '/*.*_SIZE'):
    tag, ctr = os.path.basename(d).split('.')
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'byz' in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                name = r['Kernel_Name'].replace('void ', '').replace('byz::(anonymous namespace)::', '').split('(')[0]
                per[(name, r['Dispatch_Id'])] += float(r['Counter_Value'])
        for (name, _), v in per.items():
            acc['%s/%s' % (tag, name)][ctr].append(v)
res = {}
for key, c in sorted(acc.items()):
    f, w = c['FETCH_SIZE'], c['WRITE_SIZE']
    res[key] = {'launches': len(f), 'FETCH_SIZE_KB_sum': sum(f), 'WRITE_SIZE_KB_sum': sum(w)}
    if sum(f) + sum(w) > 1e5:
        print('%-60s launches %3d  FETCH %.4g GB  WRITE %.4g GB (raw counter x 1024)' % (key, len(f), sum(f) * 1024 / 1e9, sum(w) * 1024 / 1e9))
json.dump(res, open(out + '/pmc_traffic_raw.json', 'w'), indent=1)
PY
find $OUT -name '*counter_collection.csv' -size +1M -delete
find $OUT -name '*kernel_trace.csv' -delete
echo "== sanitized run (host side under ASan + UBSan), torch-free subset; the sanitized library is built HERE (round 6: its 38 MB no longer travel with the tree)"
timeout 900 bash scripts/run_sanitized.sh python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not chunked and not from_torch and not batched_client and not device_server and not torch_device and not config3 and not config2" 2>&1 | tail -15 > $OUT/sanitized.txt; tail -5 $OUT/sanitized.txt
