#!/bin/bash
# PMC passes for one kernel family.  usage: gpu_pmc.sh <tag> <run_one args...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
if [ ! -f $OUT/../counters.txt ]; then rocprofv3 -L > $OUT/../counters.txt 2>&1; fi
i=0
DEFAULT_SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE")
if [ -n "${SETS:-}" ]; then IFS=';' read -ra USE <<< "$SETS"; else USE=("${DEFAULT_SETS[@]}"); fi
for set in "${USE[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/${RUNNER:-run_one.py} "$@" > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    if 'byz' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[k] += 1
for k in agg:
    print(k, 'dispatches', cnt[k])
    for c, v in agg[k].items(): print('   %-28s %.4g per dispatch' % (c, v / cnt[k]))
PY
done 2>&1 | tee $OUT/summary.txt
find $OUT -name '*.csv' -size +2M -delete
