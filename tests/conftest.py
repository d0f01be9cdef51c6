import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE_DIR = '/root/reference'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (only in the build container)')


@pytest.fixture(scope='session')
def golden():
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_vectors.npz')
    z = np.load(path)
    cases = {}
    for key in z.files:
        case, field = key.split('/', 1)
        cases.setdefault(case, {})[field] = z[key]
    return cases


@pytest.fixture(scope='session')
def reference_modules():
    """The unmodified reference, imported read-only; skipped where it does not exist."""
    if not os.path.isfile(os.path.join(REFERENCE_DIR, 'defences.py')):
        pytest.skip('reference checkout not present on this box')
    import importlib.util
    sys.dont_write_bytecode = True
    mods = {}
    for name in ('defences', 'malicious'):
        spec = importlib.util.spec_from_file_location('reference_' + name,
                                                      os.path.join(REFERENCE_DIR, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


@pytest.fixture(scope='session')
def eng():
    """The process-wide engine on GPU 0.  Where libbyzagg.so is missing or no MI355X is visible the GPU tests are
    skipped, not errored (a plain `pytest tests` on a CPU box must stay green)."""
    from attacking_federate_learning_amd import _native
    from attacking_federate_learning_amd.engine import EngineError, get_engine
    try:
        return get_engine()
    except (EngineError, _native.NativeLibraryMissing, ValueError, NotImplementedError) as exc:
        pytest.skip('no usable MI355X / libbyzagg here: %s' % exc)
