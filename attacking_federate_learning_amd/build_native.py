"""Builds libbyzagg.so (the HIP kernels + the C ABI) in-tree with hipcc for gfx950.

    python -m attacking_federate_learning_amd.build_native [--force]

hipcc cross-compiles without a GPU.  The shared object is written next to this file so that it
travels with the source tree to the GPU box; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
BUILD = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libbyzagg.so')
ARCH = 'gfx950'

SOURCES = ['api.hip', 'column_stats.hip', 'gram.hip', 'select.hip', 'trimmed_mean.hip', 'median_window.hip',
           'round_edges.hip', 'dedup.hip', 'gram_planes.hip', 'krum_small.hip', 'window_lean.hip', 'large_rows.hip', 'tall_select.hip']
# The sorting network only orders finite values and +/-inf padding; NaN inputs are unspecified in the
# reference as well.  Without this flag every v_min/v_max is preceded by a
# canonicalising v_max (sNaN quieting), +30% VALU work in the hot kernel.
EXTRA_FLAGS = {'trimmed_mean.hip': ['-fno-honor-nans'],
               # median_window.hip keeps its tile in registers: every loop over the register array must be
               # fully unrolled (a dynamic index would demote the array to scratch), and the staging loop of
               # the larger instantiations exceeds LLVM's default budget for `#pragma unroll`.  NaN semantics
               # stay on in this file (the padding rows are +inf; a NaN anywhere in a column makes its result NaN).
               'median_window.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'gram.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'gram_planes.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'window_lean.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'krum_small.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               # numpy arithmetic, operation by operation: no FMA contraction
               'round_edges.hip': ['-ffp-contract=off'],
               'column_stats.hip': ['-ffp-contract=off']}
COMMON_FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
                '-Wno-nan-infinity-disabled']


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; libbyzagg needs the ROCm toolchain')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    import glob
    return sorted(glob.glob(os.path.join(CSRC, '*.hpp'))) + [os.path.join(HERE, '..', 'include', 'byzagg.h'), __file__]


# The host side under AddressSanitizer + UndefinedBehaviorSanitizer: same sources, kernels untouched
# (-fno-gpu-sanitize), a second library next to the first.  It needs the sanitizer runtime in the process before it is
# loaded: scripts/run_sanitized.sh sets LD_PRELOAD and BYZ_LIBRARY.
SAN_BUILD = os.path.join(HERE, 'csrc', 'build_asan')
SAN_LIB = os.path.join(HERE, 'libbyzagg_asan.so')
SAN_FLAGS = ['-g', '-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-omit-frame-pointer', '-shared-libsan']


def _compile(src, force, sanitize=False):
    obj = os.path.join(SAN_BUILD if sanitize else BUILD, src.replace('.hip', '.o'))
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + _headers()):
        cmd = [hipcc()] + COMMON_FLAGS + (SAN_FLAGS if sanitize else []) + EXTRA_FLAGS.get(src, []) + ['-c', path, '-o', obj]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, proc.stdout, proc.stderr))
        return obj, True
    return obj, False


def sanitizer_runtime():
    """Path of the shared ASan runtime that must be LD_PRELOADed before libbyzagg_asan.so is loaded."""
    proc = subprocess.run([os.path.join(os.path.dirname(hipcc()), '..', 'lib', 'llvm', 'bin', 'clang'),
                           '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True)
    path = proc.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def build(force=False, verbose=False, sanitize=False):
    """Compile every HIP source for gfx950 and link libbyzagg.so (sanitize=True: libbyzagg_asan.so).  Returns the path."""
    out_dir, lib = (SAN_BUILD, SAN_LIB) if sanitize else (BUILD, LIB)
    os.makedirs(out_dir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        results = list(pool.map(lambda s: _compile(s, force, sanitize), SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or _stale(lib, objs):
        cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + (SAN_FLAGS if sanitize else []) + objs
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (proc.stdout, proc.stderr))
        if verbose:
            print('linked', lib)
    elif verbose:
        print('up to date:', lib)
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True, sanitize='--sanitize' in sys.argv)
