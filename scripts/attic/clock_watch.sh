#!/bin/bash
# Sample shader clock and socket power while a command runs (development aid).
# usage: bash scripts/clock_watch.sh <out-file> <command...>
OUT=$1; shift
"$@" &
PID=$!
: > $OUT
while kill -0 $PID 2>/dev/null; do
    rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT
    echo >> $OUT
done
wait $PID
