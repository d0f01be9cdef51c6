"""Runs the UNMODIFIED reference (`/root/reference/main.py`, `server.py`, `user.py`, `data_sets.py`) inside the test
process, with its `defences` / `malicious` imports resolving either to the reference's own files or to this repo's
drop-in shims (`attacking_federate_learning_amd/dropin/`).  TEST INFRASTRUCTURE ONLY.

The reference needs two packages this image lacks and a dataset it would download:
  * `tensorflow`   imported by server.py:10, never used on the lines exercised -> an empty stub module;
  * `torchvision`  data_sets.py:4 -> a stub whose `datasets.MNIST` is a seeded synthetic set of MNIST's shape
                   (1 x 28 x 28 floats, labels 0..9) and whose `transforms` do what the two used ones do to a tensor.
Nothing of the reference is copied or edited: its files are imported from where they lie.
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

REFERENCE_DIR = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN_DIR = os.path.join(ROOT, 'attacking_federate_learning_amd', 'dropin')
REFERENCE_MODULES = ('main', 'server', 'user', 'data_sets', 'defences', 'malicious', 'backdoor')


class SyntheticMNIST(torch.utils.data.Dataset):
    """Seeded stand-in for torchvision.datasets.MNIST: class-dependent blobs plus noise, so that gradients differ
    between clients and a few SGD rounds move the weights measurably."""

    def __init__(self, root, download=False, train=True, transform=None):
        n = 2000 if train else 300
        gen = torch.Generator().manual_seed(1234 if train else 4321)
        self.targets = torch.arange(n) % 10
        centres = torch.randn((10, 1, 28, 28), generator=torch.Generator().manual_seed(99))
        self.data = 0.6 * centres[self.targets] + torch.randn((n, 1, 28, 28), generator=gen)
        self.transform = transform

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        x = self.data[i]
        if self.transform is not None:
            x = self.transform(x)
        return x, int(self.targets[i])


def _stub_modules():
    tv = types.ModuleType('torchvision')
    tv.datasets = types.ModuleType('torchvision.datasets')
    tv.transforms = types.ModuleType('torchvision.transforms')
    tv.datasets.MNIST = SyntheticMNIST

    class Compose:
        def __init__(self, steps):
            self.steps = steps

        def __call__(self, x):
            for s in self.steps:
                x = s(x)
            return x

    class ToTensor:
        def __call__(self, x):
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = float(mean[0]), float(std[0])

        def __call__(self, x):
            return (x - self.mean) / self.std

    tv.transforms.Compose, tv.transforms.ToTensor, tv.transforms.Normalize = Compose, ToTensor, Normalize
    stubs = {'tensorflow': types.ModuleType('tensorflow'), 'torchvision': tv, 'torchvision.datasets': tv.datasets,
             'torchvision.transforms': tv.transforms}
    import importlib.machinery
    for name, mod in stubs.items():     # libraries probe optional dependencies with importlib.util.find_spec
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
    return stubs


@contextlib.contextmanager
def reference_imports(use_dropin):
    """sys.path / sys.modules arranged so that `import main` is the reference's, and `import defences` / `import
    malicious` are the reference's own (use_dropin=False) or this repo's shims (use_dropin=True)."""
    saved_path = list(sys.path)
    saved_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True           # /root/reference is read-only
    stubs = _stub_modules()
    scoped = REFERENCE_MODULES + tuple(stubs)
    saved_modules = {name: sys.modules.pop(name) for name in scoped if name in sys.modules}
    sys.modules.update(stubs)
    sys.path[:0] = ([DROPIN_DIR] if use_dropin else []) + [REFERENCE_DIR]
    try:
        yield
    finally:
        sys.path[:] = saved_path
        for name in scoped:                  # only what this context put there; libraries imported meanwhile stay
            sys.modules.pop(name, None)
        sys.modules.update(saved_modules)
        sys.dont_write_bytecode = saved_flag


def run_reference_main(use_dropin, defense, workdir, epochs=2, users_count=12, mal_prop=0.24, num_std=1.5, seed=7):
    """`main.main(...)` of the reference for `epochs` rounds with DriftAttack; returns the server's weights after every
    `Server.defend` call, the matrix it aggregated, and which module objects `server` and `main` bound."""
    trace = {'weights': [], 'users_grads': [], 'modules': {}}
    cwd = os.getcwd()
    os.makedirs(os.path.join(workdir, 'logs'), exist_ok=True)
    os.chdir(workdir)
    try:
        with reference_imports(use_dropin):
            torch.manual_seed(seed)
            np.random.seed(seed)
            import main as ref_main
            import server as ref_server
            inner = ref_server.Server.defend

            def traced(self, defence_method, cur_epoch):       # wraps, does not replace, server.py:86-90
                trace['users_grads'].append(self.users_grads.copy())
                inner(self, defence_method, cur_epoch)
                trace['weights'].append(self.current_weights.copy())

            ref_server.Server.defend = traced
            try:
                ref_main.main(mal_prop, num_std, defense, 'MNIST', False, 4, learning_rate=0.1, fading_rate=10000,
                              momentum=0.9, batch_size=83, users_count=users_count, epochs=epochs,
                              output=os.path.join(workdir, 'out.txt'))
            finally:
                ref_server.Server.defend = inner
            trace['modules'] = {'defences': sys.modules['defences'].__file__,
                                'malicious': sys.modules['malicious'].__file__,
                                'server': ref_server.__file__, 'main': ref_main.__file__}
    finally:
        os.chdir(cwd)
    return trace
