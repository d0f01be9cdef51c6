"""Device-resident gradient assembly (reference user.py:92, server.py:34-35, 81-83) -- SURVEY.md section 8(f).

The reference builds `usr.grads = np.concatenate([param.grad...cpu().numpy().flatten() ...])` on the host and
`Server.collect_gradients` copies it into row `idx` of the persistent host matrix `users_grads`; the defences
then need that matrix on the GPU, so N x D x 4 bytes cross PCIe every round (0.5 ms at N = 100, D = 79,510 --
fifty times the aggregation itself).  `GradientMatrix` is that persistent matrix kept on the GPU: clients whose
backward pass ran on the GPU write their per-parameter gradients straight into their row (one kernel, no host
visit); clients that still hand over a numpy vector cost one row copy, as in the reference.
"""
import numpy as np

from .engine import get_engine


class GradientMatrix:
    """`Server.users_grads` (server.py:35) on the device: `n_users` rows of `n_params` fp32 values."""

    def __init__(self, n_users, n_params, engine=None, torch_device=None):
        self.engine = engine or get_engine()
        self.shape = (int(n_users), int(n_params))
        if torch_device is not None:
            import torch
            self.data = torch.empty(self.shape, dtype=torch.float32, device=torch_device)
        else:
            self.data = self.engine.empty(self.shape, np.float32)

    def set_row(self, idx, grads):
        """`self.users_grads[idx, :] = usr.grads` (server.py:83).  `grads`: flat host vector, or the list of the
        client's per-parameter device gradients in parameter order (user.py:92's concatenation happens on the GPU)."""
        self.engine.assemble_row(self.data, idx, grads)

    def collect_gradients(self, users):
        """server.py:81-83.  Clients whose gradients are lists of device tensors (one per parameter) are written by ONE
        launch per run of such clients (`byz_assemble_rows_dev`: the pointer table of all of them goes to the device once)
        instead of one launch per client -- 0.99 ms per round of 100 clients that way, launch bound; clients that hand over a
        flat host vector cost one row copy each, as in the reference."""
        run_start, run = 0, []
        for idx, usr in enumerate(users):
            grads = usr.grads
            if isinstance(grads, (list, tuple)):
                if not run:
                    run_start = idx
                run.append(grads)
                continue
            if run:
                self.engine.assemble_rows(self.data, run_start, run)
                run = []
            self.set_row(idx, grads)
        if run:
            self.engine.assemble_rows(self.data, run_start, run)

    def set_all(self, batched_grads):
        """All rows from a batched client step: one device tensor (n_users, *shape) per model parameter."""
        self.engine.assemble_columns(self.data, batched_grads)

    def numpy(self):
        return self.data.cpu().numpy() if hasattr(self.data, 'cpu') else self.data.numpy()
