#!/bin/bash
# HBM traffic of the dominant kernels from PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE, as
# MI355X_MICROARCH.md prescribes), written to gpurun_out/traffic/hbm_traffic.json.
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # tag counter cmd...
  tag=$1; ctr=$2; shift 2
  timeout 240 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$tag.$ctr -o p --output-format csv -- "$@" > $OUT/$tag.$ctr.log 2>&1
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  run c4 $ctr python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0
  run c3 $ctr python $GRAFT_REPO_ROOT/bench.py --workload c3 --no-cpu-baseline --steps 3 --warmup 1
  run c2 $ctr python $GRAFT_REPO_ROOT/bench.py --workload c2 --no-cpu-baseline --steps 3 --warmup 1
  run attack $ctr python $GRAFT_REPO_ROOT/bench.py --workload attack --params 1000000 --no-cpu-baseline --steps 3 --warmup 1
done
python3 - <<'PY'
import csv, glob, json, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/traffic'
names = {'gram_tile_kernel': 'gram_tile', 'gram_planes_kernel': 'gram_tile', 'median_window_kernel': 'trimmed_mean',
         'column_partial_kernel': 'column_stats'}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(out + '/*.*_SIZE'):
    tag, ctr = os.path.basename(d).split('.')
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            for k, short in names.items():
                if k in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                    per[(short, r['Dispatch_Id'])] += float(r['Counter_Value'])
        for (short, _), v in per.items():
            acc['%s/%s' % (tag, short)][ctr].append(v)
res = {}
for key, c in acc.items():
    fetch = sum(c['FETCH_SIZE']) / max(len(c['FETCH_SIZE']), 1)
    write = sum(c['WRITE_SIZE']) / max(len(c['WRITE_SIZE']), 1)
    res[key] = {'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write,
                # gfx950: FETCH_SIZE reports half the bytes of a wide (16 B/lane) streaming read -> doubled
                'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024.0,
                'launches_sampled': len(c['FETCH_SIZE']),
                'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH + WRITE) * 1024'}
json.dump(res, open(out + '/hbm_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name '*.csv' -size +1M -delete
