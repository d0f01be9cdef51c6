#!/bin/bash
# Runs a command against the AddressSanitizer + UBSan build of the host library:
#   scripts/run_sanitized.sh python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or small_krum"
# The kernels are the same objects as in libbyzagg.so; what is checked is the C++ host side (workspace growth, argument
# handling, staging copies, launch arithmetic).
set -e
cd "$(dirname "$0")/.."
python -m attacking_federate_learning_amd.build_native --sanitize > /dev/null
RT=$(python -c "from attacking_federate_learning_amd import build_native as b; print(b.sanitizer_runtime() or '')")
[ -n "$RT" ] || { echo "no shared ASan runtime in this ROCm installation" >&2; exit 1; }
export LD_PRELOAD="$RT${LD_PRELOAD:+:$LD_PRELOAD}"
export BYZ_LIBRARY="$PWD/attacking_federate_learning_amd/libbyzagg_asan.so"
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:${ASAN_OPTIONS}"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:${UBSAN_OPTIONS}"
exec "$@"
