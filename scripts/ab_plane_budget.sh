cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for mb in 16384 8192 4096; do
  BYZ_GRAM_PLANE_MB=$mb python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-sharded-w1 --no-north-star --detail-file '' 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readlines()[-1])
k = l.get('kernels_ms') or {}
print('plane_mb=$mb ms_per_step %.1f gram %.1f split %.1f launches %s' % (l['ms_per_step'], k.get('gram_tile', -1), k.get('plane_split', -1), l['roofline'].get('launches_per_step')))
"
done; done
