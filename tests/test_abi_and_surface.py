"""CPU checks of the boundary: the shared library exports what include/byzagg.h declares, the drop-in modules
expose the reference's names and signatures, and the product refuses to run without its HIP path."""
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'byzagg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(byz_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from attacking_federate_learning_amd import _native, build_native
    build_native.build()
    lib = _native.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_native.EXPORTED_SYMBOLS) == names       # the ctypes table covers the whole header
    assert lib.byz_abi_version() == 1
    assert lib.byz_kernel_name(1) == b'gram_tile'


def declared_prototypes():
    """name -> list of C parameter types, parsed from include/byzagg.h."""
    text = open(os.path.join(ROOT, 'include', 'byzagg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r'\b(int|const char\*|void)\s+(byz_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        args = ' '.join(args.split())
        protos[name] = [] if args in ('', 'void') else [a.strip() for a in args.split(',')]
    return protos


def test_ctypes_table_matches_the_header_argument_by_argument():
    """A ctypes prototype that disagrees with the header corrupts a call silently: compare count and kind."""
    import ctypes
    from attacking_federate_learning_amd import _native
    protos = declared_prototypes()
    assert sorted(protos) == sorted(_native.EXPORTED_SYMBOLS)

    def kind(c_type):   # how the C declaration must look for this ctypes argument type
        if c_type in (ctypes.c_int64,):
            return 'int64'
        if c_type in (ctypes.c_int, ctypes.c_int32):
            return 'int'
        if c_type is ctypes.c_float:
            return 'float'
        return 'pointer'

    def c_kind(decl):
        base = decl.rsplit(' ', 1)[0] if ' ' in decl else decl
        if '*' in decl or base.endswith('_fn'):    # (a callback typedef, e.g. byz_allreduce_f64_fn, is a function pointer)
            return 'pointer'
        if 'int64_t' in base:
            return 'int64'
        if 'float' in base:
            return 'float'
        if re.search(r'\bint(32_t)?\b', base):
            return 'int'
        raise AssertionError('unrecognised parameter %r' % decl)

    for name, argtypes in _native._PROTOTYPES.items():
        decl = protos[name]
        assert len(decl) == len(argtypes), (name, decl, argtypes)
        for position, (d, a) in enumerate(zip(decl, argtypes)):
            assert c_kind(d) == kind(a), (name, position, d, a)


def test_limits_are_reported_without_a_gpu():
    import ctypes
    from attacking_federate_learning_amd import _native
    lib = _native.load()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert lib.byz_limits(ctypes.byref(a), ctypes.byref(b)) == 0
    assert a.value >= 10000 and b.value >= 10000    # config 5: N = 10000 clients for every defence of defences.defend


def test_no_cpu_fallback():
    """Without a GPU the product path fails loudly instead of computing somewhere else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from attacking_federate_learning_amd import defences, malicious
    g = np.zeros((5, 8), dtype=np.float32)
    for name in ('NoDefense', 'Krum', 'TrimmedMean', 'Bulyan'):
        with pytest.raises(RuntimeError):
            defences.defend[name](g, 5, 0)

    class U:
        grads, original_params, learning_rate = g[0], None, None
    with pytest.raises(RuntimeError):
        malicious.DriftAttack(1.5).attack([U()])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'attacking_federate_learning_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle/' not in src or f == 'sharded.py' or 'tests/' in src, f


def test_drop_in_surface_matches_the_reference(reference_modules):
    from attacking_federate_learning_amd import defences, malicious
    ref_d, ref_m = reference_modules['defences'], reference_modules['malicious']
    for name in ('no_defense', '_krum_create_distances', 'krum', 'trimmed_mean', 'bulyan'):
        assert str(inspect.signature(getattr(defences, name))) == str(inspect.signature(getattr(ref_d, name))), name
    assert list(defences.defend) == list(ref_d.defend)
    for attr in ('NoDefense', 'Krum', 'TrimmedMean', 'Bulyan'):
        assert getattr(defences.DefenseTypes, attr) == getattr(ref_d.DefenseTypes, attr)
    for cls in ('Attack', 'DriftAttack'):
        for meth in ('__init__', 'attack'):
            assert str(inspect.signature(getattr(getattr(malicious, cls), meth))) == \
                str(inspect.signature(getattr(getattr(ref_m, cls), meth)))
    assert str(inspect.signature(malicious.DriftAttack._attack_grads)) == \
        str(inspect.signature(ref_m.DriftAttack._attack_grads))
    assert issubclass(malicious.DriftAttack, malicious.Attack)
    att = malicious.Attack(1.5)
    assert (att.num_std, att.grads_mean, att.grads_stdev) == (1.5, None, None)


def test_dropin_shims_resolve():
    import importlib.util
    for name in ('defences', 'malicious'):
        path = os.path.join(ROOT, 'attacking_federate_learning_amd', 'dropin', name + '.py')
        spec = importlib.util.spec_from_file_location('shim_' + name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if name == 'defences':
            assert callable(mod.defend['Krum']) and mod.DefenseTypes.Bulyan == 'Bulyan'
        else:
            assert issubclass(mod.DriftAttack, mod.Attack)


def test_host_library_under_address_and_ub_sanitizers(tmp_path):
    """the C++ host side of libbyzagg built with -fsanitize=address,undefined (kernels untouched).
    Without a GPU only the entry points that never reach a kernel run -- context creation failing on the missing device,
    argument checks, the name tables -- but they run under the sanitizers, and the sanitized library must export the
    whole ABI.  With a GPU the same library runs the parity tests: scripts/run_sanitized.sh."""
    import subprocess
    import sys
    from attacking_federate_learning_amd import build_native
    runtime = build_native.sanitizer_runtime()
    if runtime is None:
        pytest.skip('this ROCm installation has no shared ASan runtime')
    lib = build_native.build(sanitize=True)
    snippet = tmp_path / 'calls.py'
    snippet.write_text(
        'import ctypes, sys\n'
        'sys.path.insert(0, %r)\n'
        'from attacking_federate_learning_amd import _native\n'
        'lib = _native.load()          # resolves every prototype of include/byzagg.h\n'
        'assert lib.byz_abi_version() == 1\n'
        'names = [lib.byz_kernel_name(k) for k in range(-2, 14)]\n'
        'a, b = ctypes.c_int64(0), ctypes.c_int64(0)\n'
        'assert lib.byz_limits(ctypes.byref(a), ctypes.byref(b)) == 0 and a.value >= 10000\n'
        'ctx = ctypes.c_void_p()\n'
        'rc = lib.byz_ctx_create(0, ctypes.byref(ctx))\n'
        'if rc == 0:\n'
        '    assert lib.byz_ctx_reserve(ctx, 100, 1000) == 0\n'
        '    assert lib.byz_ctx_reserve(ctx, -1, 5) == _native.E_INVALID\n'
        '    lib.byz_ctx_destroy(ctx)\n'
        'else:\n'
        '    assert rc == _native.E_HIP and _native.last_error()\n'
        'assert lib.byz_ctx_reserve(None, 10, 10) == _native.E_INVALID\n'
        'rows = ctypes.c_int64(0)\n'
        'assert lib.byz_bulyan_rescored(None, ctypes.byref(rows)) == _native.E_INVALID\n'
        'assert lib.byz_timing_reset(None) == _native.E_INVALID\n'
        'lib.byz_ctx_destroy(None)\n'
        'print("sanitized calls done")\n' % ROOT)
    env = dict(os.environ, LD_PRELOAD=runtime, BYZ_LIBRARY=lib,
               ASAN_OPTIONS='detect_leaks=0:protect_shadow_gap=0', UBSAN_OPTIONS='print_stacktrace=1')
    proc = subprocess.run([sys.executable, str(snippet)], capture_output=True, text=True, env=env, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert 'sanitized calls done' in proc.stdout
    assert 'AddressSanitizer' not in proc.stderr and 'runtime error' not in proc.stderr, proc.stderr[-2000:]


def test_the_header_is_plain_c_and_the_c_host_example_compiles():
    """include/byzagg.h is a C header (not C++): it must pass `gcc -std=c99 -pedantic`; and the C host of INTEGRATION.md 5b
    (examples/shard_columns.c, which needs HIP's and RCCL's headers) must at least compile here, without warnings."""
    import subprocess
    probe = '#include "byzagg.h"\nint main(void) { return byz_abi_version() == BYZ_ABI_VERSION ? 0 : 1; }\n'
    proc = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-fsyntax-only', '-I', os.path.join(ROOT, 'include'),
                           '-x', 'c', '-'], input=probe, capture_output=True, text=True)
    assert proc.returncode == 0 and not proc.stderr.strip(), proc.stderr
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    if not os.path.isfile(os.path.join(rocm, 'include', 'rccl', 'rccl.h')):
        pytest.skip('rccl/rccl.h not installed')
    proc = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-fsyntax-only', '-D__HIP_PLATFORM_AMD__',
                           '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(rocm, 'include'),
                           os.path.join(ROOT, 'examples', 'shard_columns.c')], capture_output=True, text=True)
    assert proc.returncode == 0 and not proc.stderr.strip(), proc.stderr[-2000:]
