#!/bin/bash
# Round 6, visit x: claimed runs with a soft round gate (the next round starts when all but `slack` workgroups of the last are done)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06x
mkdir -p $O
export BYZ_GRAM_CLAIM=1
REPS=3 CALLS=3 timeout 600 python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_SLACK=0 BYZ_GRAM_SLACK=2 BYZ_GRAM_SLACK=4 BYZ_GRAM_SLACK=8 BYZ_GRAM_SLACK=16 BYZ_GRAM_ROUND=0 BYZ_GRAM_CLAIM=0 2>&1 | grep rep > $O/slack_ab_n4000.txt
REPS=2 CALLS=2 timeout 600 python scripts/gram_span_ab.py 10000 401408 BYZ_GRAM_SLACK=0 BYZ_GRAM_SLACK=2 BYZ_GRAM_SLACK=4 BYZ_GRAM_SLACK=8 BYZ_GRAM_ROUND=0 2>&1 | grep rep > $O/slack_ab_n10000.txt
cat $O/slack_ab_n4000.txt $O/slack_ab_n10000.txt
