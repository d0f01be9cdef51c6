"""CPU restatement of the client step (reference user.py:21-29, 76-92).  TEST INFRASTRUCTURE ONLY.

One client per call, exactly the reference's sequence: load the server's weight row into the network's
parameters, one forward pass over the client's batch, NLL loss on the network's log-softmax output, one backward
pass, and the per-parameter gradients flattened and concatenated in parameter order.  Pinned against the
reference's own `User.step` (tests/test_oracle_vs_reference.py, and the `clients_*` golden case).
"""
import functools

import numpy as np
import torch


def row_into_parameters(row, parameters):
    """user.py:21-29 -- consecutive slices of the flat weight row, reshaped, in parameter order."""
    offset = 0
    for param in parameters:
        size = functools.reduce(lambda a, b: a * b, param.shape)
        param.data[:] = torch.from_numpy(np.asarray(row[offset:offset + size]).reshape(tuple(param.shape)))
        offset += size


def client_gradient(net, current_params, data, target, flatten_input=True):
    """user.py:85-92 (step) and :68-80 (train).  `flatten_input` is the MNIST branch (data.view(-1, 28 * 28))."""
    row_into_parameters(current_params, net.parameters())
    if flatten_input:
        data = data.view(-1, 28 * 28)
    net.zero_grad()                                   # optimizer.zero_grad() on a fresh optimizer over net.parameters()
    loss = torch.nn.NLLLoss()(net(data), target)      # user.py:36, 78-79
    loss.backward()
    return np.concatenate([p.grad.data.cpu().numpy().flatten() for p in net.parameters()])   # user.py:92


def all_client_gradients(net, current_params, data, target, flatten_input=True):
    """server.py:54-56 then 81-83: every client steps from the same weights; row idx = client idx."""
    rows = [client_gradient(net, current_params, data[c], target[c], flatten_input) for c in range(len(data))]
    return np.stack(rows)


class MnistNet(torch.nn.Module):
    """The reference's MNIST network (data_sets.py:13-23): 784 -> 100 -> 10, ReLU, log-softmax output.
    Parameter order fc1.weight, fc1.bias, fc2.weight, fc2.bias = 79,510 values."""

    def __init__(self):
        super().__init__()
        self.fc1 = torch.nn.Linear(28 * 28, 100)
        self.fc2 = torch.nn.Linear(100, 10)

    def forward(self, x):
        return torch.log_softmax(self.fc2(torch.relu(self.fc1(x))), dim=1)
