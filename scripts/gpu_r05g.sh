#!/bin/bash
# Round 5, visit g: counters.  The Gram tile kernel after the block masks and the deferred slab update (issued MFMAs against
# 3 N^2 D / 32768, matrix-pipe busy cycles, LDS and vector instruction counts) at N = 4000 and N = 10,000, and the
# register-resident attack kernel.  PMC passes only (kernel trace, no other tracing).
set -u
export TMPDIR=/tmp
export ITERS=2
SETS="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
SETS="$SETS" bash scripts/gpu_pmc.sh r05g_gram_n4000 gram 4000 1000000 > /dev/null 2>&1
SETS="$SETS" bash scripts/gpu_pmc.sh r05g_gram_n10000 gram 10000 262144 > /dev/null 2>&1
SETS="$SETS;SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" bash scripts/gpu_pmc.sh r05g_attack attack 2400 1000000 > /dev/null 2>&1
for t in r05g_gram_n4000 r05g_gram_n10000 r05g_attack; do echo "== $t"; grep -A14 "gram_planes_kernel\|column_resident_kernel\|chunk_reduce" gpurun_out/$t/summary.txt | head -60; done
