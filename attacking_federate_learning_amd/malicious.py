"""Drop-in replacement for the reference's `malicious` module (reference malicious.py:1-36).

`Attack.attack(users)` keeps the reference's contract: it reads `usr.grads` of every malicious user, stores
`grads_mean` / `grads_stdev` on the attacker, returns early when `num_std == 0`, otherwise calls the
overridable hook `_attack_grads(grads_mean, grads_stdev, original_params, learning_rate)` with host arrays
and rebinds every `usr.grads` to the ONE array the hook returned (reference malicious.py:26-27).
`backdoor.BackdoorAttack` subclasses `Attack` and overrides only the hook, which keeps working.

The column statistics run in libbyzagg's column kernel; for `DriftAttack` the drift itself is fused into
that kernel, so the m x D matrix is read once.
"""
import numpy as np

from .engine import get_engine


class Attack(object):
    def __init__(self, num_std):
        self.num_std = num_std
        self.grads_mean = None
        self.grads_stdev = None

    def _statistics(self, users, num_std):
        rows = np.stack([np.asarray(usr.grads, dtype=np.float32) for usr in users])
        return get_engine().drift_attack(rows, num_std)

    def attack(self, users):
        if len(users) == 0:
            return

        drift, self.grads_mean, self.grads_stdev = self._statistics(users, self.num_std)

        if self.num_std == 0:
            return

        mal_grads = self._attack_grads(self.grads_mean, self.grads_stdev, users[0].original_params,
                                       users[0].learning_rate)

        for usr in users:
            usr.grads = mal_grads


class DriftAttack(Attack):
    def __init__(self, num_std):
        super(DriftAttack, self).__init__(num_std)
        self._fused = None

    def attack(self, users):
        if len(users) == 0:
            return
        # one kernel produces mean, std and mean - z*std; the hook below hands the fused vector back
        self._fused, self.grads_mean, self.grads_stdev = self._statistics(users, self.num_std)
        if self.num_std == 0:
            self._fused = None
            return
        mal_grads = self._attack_grads(self.grads_mean, self.grads_stdev, users[0].original_params,
                                       users[0].learning_rate)
        self._fused = None
        for usr in users:
            usr.grads = mal_grads

    def _attack_grads(self, grads_mean, grads_stdev, original_params, learning_rate):
        """mean[:] -= num_std * std[:], in place, returns `grads_mean` (reference malicious.py:34-36)."""
        if self._fused is not None and grads_mean is self.grads_mean:
            grads_mean[:] = self._fused
        else:  # called directly with arbitrary vectors
            grads_mean[:] = get_engine().drift_axpy_host(grads_mean, grads_stdev, self.num_std)
        return grads_mean
