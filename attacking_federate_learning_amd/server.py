"""The server's round state on the GPU (reference server.py:33-36, 54-56, 81-90) -- SURVEY.md 8(a) rows a9/a10.

The reference's `Server` keeps `current_weights`, `velocity` and the N x D matrix `users_grads` as host numpy
arrays; every round the clients' gradients are copied into the matrix row by row (collect_gradients), the chosen
defence reduces it to one vector (defend) and a momentum step moves the weights.  `DeviceServer` keeps those three
arrays on the MI355X and runs the same three steps there, so a round never crosses PCIe:

    collect_gradients(users)        server.py:81-83   rows written by libbyzagg (host vectors or device tensors)
    collect_batched(net, x, y)      server.py:54-56 + user.py:76-92 for every client at once (clients.py)
    attack(attacker_rows, num_std)  main.py:68 -> malicious.py:10-36 on the first rows, in place
    defend(defence_method)          server.py:86-90   defences.defend[...] on the device matrix + fused momentum step

Only what is on the aggregation path is mirrored: evaluation, checkpoints, logging and data loading stay the
reference's own code.
"""
import numpy as np

from . import defences
from .assembly import GradientMatrix
from .engine import get_engine


class DeviceServer:
    def __init__(self, n_users, current_weights, mal_prop, learning_rate, momentum, torch_device='cuda', engine=None):
        import torch
        self.engine = engine or get_engine()
        self.n_users = int(n_users)
        self.mal_prop, self.learning_rate, self.momentum = mal_prop, learning_rate, momentum
        # server.py:33-36: weights as one flat fp32 row, the gradient matrix, a zero velocity
        self.current_weights = torch.as_tensor(np.asarray(current_weights, dtype=np.float32)).to(torch_device).clone()
        self.users_grads = GradientMatrix(self.n_users, self.current_weights.numel(), engine=self.engine,
                                          torch_device=torch_device)
        self.velocity = torch.zeros_like(self.current_weights)

    # ---- server.py:81-83 ---------------------------------------------------------------------------
    def collect_gradients(self, users):
        self.users_grads.collect_gradients(users)

    # ---- server.py:54-56 + 81-83 with every client's step batched (user.py:76-92) --------------------
    def collect_batched(self, net, data, target):
        from .clients import collect_batched
        collect_batched(self.users_grads, net, self.current_weights, data, target)

    # ---- main.py:68: the first rows are the malicious clients (main.py:28) ---------------------------
    def attack(self, n_malicious, num_std):
        """A Little Is Enough on rows 0 .. n_malicious-1, in place (malicious.py:10-36); returns mean, std."""
        if n_malicious <= 0:
            return None, None
        rows = self.users_grads.data[:n_malicious]
        _, mean, std = self.engine.drift_attack(rows, num_std, write_back=(num_std != 0))
        return mean, std

    # ---- server.py:86-90 ---------------------------------------------------------------------------
    def defend(self, defence_method, cur_epoch=None):
        current_grads = defences.defend[defence_method](self.users_grads.data, self.n_users,
                                                        int(self.n_users * self.mal_prop))
        # velocity = momentum * velocity - learning_rate * current_grads ; current_weights += velocity
        self.engine.server_update(self.current_weights, self.velocity, current_grads, self.momentum,
                                  self.learning_rate)
        return current_grads
