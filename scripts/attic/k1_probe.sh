cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 3; do echo -n "dbg=$dbg "; BYZ_KRUM_SMALL_DBG=$dbg BYZ_KRUM_SMALL_SKIP=2 timeout 120 python scripts/c2_rounds.py 79510 2000 2>&1 | tail -1; done
