"""ctypes binding of libbyzagg.so (include/byzagg.h).  No fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BYZ_LIBRARY: another build of the same ABI (the sanitized one, build_native.py --sanitize)
LIB_PATH = os.environ.get('BYZ_LIBRARY') or os.path.join(_HERE, 'libbyzagg.so')

OK, E_INVALID, E_PRECONDITION, E_HIP, E_UNSUPPORTED, E_NO_WINNER, E_COLLECTIVE = 0, -1, -2, -3, -4, -5, -6

KERNELS = ('column_stats', 'gram_tile', 'gram_reduce', 'distances', 'row_sort', 'krum_argmin',
           'bulyan_loop', 'trimmed_mean', 'misc', 'plane_split')

c_i64, c_i32, c_int, c_f32, c_vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_P = ctypes.POINTER

# name -> argument types (everything returns int unless listed in _RESTYPES)
_PROTOTYPES = {
    'byz_abi_version': [],
    'byz_last_error': [],
    'byz_ctx_create': [c_int, _P(c_vp)],
    'byz_ctx_destroy': [c_vp],
    'byz_ctx_reserve': [c_vp, c_i64, c_i64],
    'byz_ctx_device': [c_vp],
    'byz_limits': [_P(c_i64), _P(c_i64)],
    'byz_malloc': [c_vp, c_i64, _P(c_vp)],
    'byz_free': [c_vp, c_vp],
    'byz_upload': [c_vp, c_vp, c_vp, c_i64, c_vp],
    'byz_download': [c_vp, c_vp, c_vp, c_i64, c_vp],
    'byz_upload_2d': [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp],
    'byz_stream_sync': [c_vp, c_vp],
    'byz_no_defense_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp],
    'byz_pairwise_distances_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp],
    'byz_gram_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp],
    'byz_gram_share_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_vp, c_vp],
    'byz_gram_share_add_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_vp, c_vp],
    'byz_distances_from_gram_dev': [c_vp, c_vp, c_i64, c_vp, c_vp],
    'byz_near_pairs_count': [c_vp, _P(c_i64), c_vp],
    'byz_near_pairs_sqdist_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    'byz_near_pairs_apply_dev': [c_vp, c_vp, c_i64, c_vp, c_vp],
    'byz_ctx_check': [c_vp, c_vp],
    'byz_bulyan_rescored': [c_vp, _P(c_i64)],
    'byz_krum_select_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, _P(c_i32), c_vp, c_vp],
    'byz_krum_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, _P(c_i32), c_vp],
    'byz_trimmed_mean_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp],
    'byz_trimmed_mean_redone': [c_vp, _P(c_i64), c_vp],
    'byz_bulyan_select_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp],
    'byz_krum_bulyan_select_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, _P(c_i32), c_vp, c_vp],
    'byz_bulyan_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    # (the byz_allreduce_f64_fn callback and its `user` word are passed as plain pointers: ALLREDUCE_F64_FN builds the callback)
    'byz_pairwise_distances_sharded_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp],
    'byz_krum_sharded_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, _P(c_i32), c_vp],
    'byz_bulyan_sharded_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    'byz_drift_attack_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, c_int, c_vp],
    'byz_column_chain_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp],
    'byz_column_finish_dev': [c_vp, c_vp, c_vp, c_i64, c_f32, c_i64, c_vp, c_vp, c_vp, c_vp],
    'byz_drift_axpy_dev': [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp],
    'byz_server_update_dev': [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp],
    'byz_backdoor_initial_params_dev': [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp],
    'byz_backdoor_clip_dev': [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp],
    'byz_assemble_row_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    'byz_assemble_rows_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    'byz_assemble_rows_again_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp],
    'byz_assemble_columns_dev': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    'byz_assemble_row_host': [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp],
    'byz_defend_host': [c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_vp],
    'byz_pairwise_distances_host': [c_vp, c_vp, c_i64, c_i64, c_vp],
    'byz_krum_select_host': [c_vp, c_vp, c_i64, c_i64, c_i64, _P(c_i32)],
    'byz_drift_attack_host': [c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp],
    'byz_timing_enable': [c_vp, c_int],
    'byz_timing_reset': [c_vp],
    'byz_timing_read': [c_vp, c_int, _P(ctypes.c_double), _P(c_i64)],
    'byz_kernel_name': [c_int],
    'byz_selftest_lane_exchange_dev': [c_vp, c_vp, _P(c_i32), c_vp],
}
_RESTYPES = {'byz_last_error': ctypes.c_char_p, 'byz_kernel_name': ctypes.c_char_p, 'byz_ctx_destroy': None}

# int (*byz_allreduce_f64_fn)(void* user, double* buf_dev, int64_t count, void* stream)
ALLREDUCE_F64_FN = ctypes.CFUNCTYPE(c_int, c_vp, c_vp, c_i64, c_vp)

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).  If
    libbyzagg pulled in /opt/rocm's copies first, a later `import torch` would mix the two sets and find no
    GPU.  So when torch is installed its copies are loaded first (without importing torch), and both torch
    and libbyzagg bind to them; device pointers and streams are then interchangeable.
    """
    if os.environ.get('BYZ_SYSTEM_HIP') == '1':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    for name in ('libhsa-runtime64.so', 'libamd_comgr.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def load():
    """Load libbyzagg.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            'libbyzagg.so is not built (%s).  Run `python -m attacking_federate_learning_amd.build_native` '
            '(needs hipcc); this package has no CPU fallback.' % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here means the library and the header disagree
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.byz_abi_version() != 1:
        raise RuntimeError('libbyzagg ABI version mismatch')
    _lib = lib
    return lib


def last_error():
    msg = load().byz_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''
