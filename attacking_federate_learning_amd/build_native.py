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
           'round_edges.hip', 'dedup.hip', 'gram_planes.hip', 'window_rows.hip', 'krum_small.hip']
# The sorting network only orders finite values and +/-inf padding; NaN inputs are unspecified in the
# reference as well.  Without this flag every v_min/v_max is preceded by a
# canonicalising v_max (sNaN quieting), +30% VALU work in the hot kernel.
EXTRA_FLAGS = {'trimmed_mean.hip': ['-fno-honor-nans'],
               # median_window.hip keeps its tile in registers: every loop over the register array must be
               # fully unrolled (a dynamic index would demote the array to scratch), and the staging loop of
               # the larger instantiations exceeds LLVM's default budget for `#pragma unroll`.  NaN semantics
               # stay on in this file: the padding rows are NaNs.
               'median_window.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'gram.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'gram_planes.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'window_rows.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               'krum_small.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'],
               # numpy arithmetic, operation by operation: no FMA contraction
               'round_edges.hip': ['-ffp-contract=off']}
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
    return [os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'lane_exchange.hpp'), os.path.join(HERE, '..', 'include', 'byzagg.h'), __file__]


def _compile(src, force):
    obj = os.path.join(BUILD, src.replace('.hip', '.o'))
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + _headers()):
        cmd = [hipcc()] + COMMON_FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', path, '-o', obj]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, proc.stdout, proc.stderr))
        return obj, True
    return obj, False


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libbyzagg.so.  Returns the library path."""
    os.makedirs(BUILD, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        results = list(pool.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or _stale(LIB, objs):
        cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (proc.stdout, proc.stderr))
        if verbose:
            print('linked', LIB)
    elif verbose:
        print('up to date:', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
