#!/bin/bash
# HBM-side traffic of the c4 round's kernels (PMC FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md says)
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/c4.$ctr -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 > $OUT/c4.$ctr.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/traffic'
names = {'gram_planes_kernel': 'gram_tile', 'plane_split_f16_kernel': 'plane_split', 'window_rows_kernel': 'trimmed_mean',
         'median_window_kernel': 'trimmed_mean_redo', 'bulyan_grid_kernel': 'bulyan_loop'}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(out + '/c4.*_SIZE'):
    ctr = os.path.basename(d).split('.')[1]
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            for k, short in names.items():
                if k in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                    per[(short, r['Dispatch_Id'])] += float(r['Counter_Value'])
        for (short, _), v in per.items():
            acc[short][ctr].append(v)
res = {}
for key, c in acc.items():
    res[key] = {'FETCH_SIZE_KB': c['FETCH_SIZE'], 'WRITE_SIZE_KB': c['WRITE_SIZE']}
json.dump(res, open(out + '/c4_raw.json', 'w'), indent=1)
for key, c in acc.items():
    f, w = c['FETCH_SIZE'], c['WRITE_SIZE']
    print(key, 'launches', len(f), 'FETCH KB avg %.4g sum %.4g' % (sum(f) / max(len(f), 1), sum(f)), 'WRITE KB avg %.4g sum %.4g' % (sum(w) / max(len(w), 1), sum(w)))
PY
find $OUT -name '*.csv' -size +1M -delete
