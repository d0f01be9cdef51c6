"""The clients' gradient step, batched (reference user.py:76-92, server.py:54-56) -- SURVEY.md section 8(f) rank 3.

In the reference every client copies the server's weights into its own network, runs one forward/backward pass
over its own mini-batch and hands back the flattened gradient (N sequential passes, N host round trips).  All
clients start from the SAME weights, so the N passes are one batched pass: `torch.func.vmap` over the client axis
of `grad(loss)`, on the GPU, and the per-parameter gradients (N, *shape) go straight into the rows of the
device-resident `GradientMatrix` (one `byz_assemble_columns_dev` launch).  Nothing visits the host.

The network is the caller's (`data_sets.MnistNet` / `Cifar10Net` in the reference): any `torch.nn.Module` whose
forward returns log-probabilities, as the reference's do.  This is host-side PyTorch-ROCm plumbing around the
aggregation path, not part of it.
"""


def weights_to_parameters(net, current_weights):
    """user.py:21-29 without the copy: views of the flat weight row, one per parameter, in parameter order."""
    import torch
    first = next(net.parameters())
    flat = torch.as_tensor(current_weights, dtype=torch.float32, device=first.device).reshape(-1)
    params, offset = {}, 0
    for name, p in net.named_parameters():
        n = p.numel()
        params[name] = flat[offset:offset + n].view(p.shape)
        offset += n
    if offset != flat.numel():
        raise ValueError('the weight row holds %d values, the network %d' % (flat.numel(), offset))
    return params


def per_client_gradients(net, current_weights, data, target):
    """Gradients of every client's mean NLL loss (user.py:36, 78-80) at the shared weights.

    data: (N, B, ...) -- client c's mini-batch is data[c]; target: (N, B) class indices.
    Returns one tensor (N, *param.shape) per parameter, in parameter order (the order user.py:92 concatenates)."""
    import torch
    from torch.func import functional_call, grad, vmap
    params = weights_to_parameters(net, current_weights)
    buffers = dict(net.named_buffers())

    def client_loss(p, x, y):
        return torch.nn.functional.nll_loss(functional_call(net, (p, buffers), (x,)), y)

    grads = vmap(grad(client_loss), in_dims=(None, 0, 0))(params, data, target)
    return [grads[name] for name, _ in net.named_parameters()]


def collect_batched(matrix, net, current_weights, data, target):
    """dispatch_weights + collect_gradients (server.py:54-56, 81-83) in one batched step: fills every row of the
    device-resident `GradientMatrix`."""
    matrix.set_all(per_client_gradients(net, current_weights, data, target))
    return matrix
