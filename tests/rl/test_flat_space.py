"""FlatParameterSpace: parameters and gradients as views of flat buffers; gradients gathered after backward."""

from __future__ import annotations

import torch

from alphatriangle_amd.ops.optim import FlatParameterSpace


def _net() -> torch.nn.Module:
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))


def test_gather_grads_matches_accumulating_into_the_views():
    net = _net()
    ref = _net()
    space = FlatParameterSpace(net)
    x = torch.randn(7, 6)
    # reference: plain autograd
    ref(x).square().sum().backward()
    # flat space: detach the views, let autograd keep its own tensors, gather once
    space.release_grads()
    assert all(p.grad is None for p in space.params)
    net(x).square().sum().backward()
    space.gather_grads()
    views = space.view_like_params(space.flat_grad)
    for p, v, q in zip(space.params, views, ref.parameters(), strict=True):
        assert p.grad.data_ptr() == v.data_ptr()  # views are re-attached
        assert torch.allclose(p.grad, q.grad, atol=1e-6)


def test_gather_grads_zeroes_parameters_that_took_no_part():
    net = _net()
    space = FlatParameterSpace(net)
    space.flat_grad.fill_(5.0)  # stale values from an earlier step
    space.release_grads()
    x = torch.randn(4, 6)
    net[0](x).sum().backward()  # only the first layer takes part
    space.gather_grads()
    used = {id(p) for p in net[0].parameters()}
    for p in space.params:
        if id(p) in used:
            assert p.grad.abs().sum() > 0
        else:
            assert torch.count_nonzero(p.grad) == 0


def test_parameters_stay_views_of_the_flat_buffer_after_gather():
    net = _net()
    space = FlatParameterSpace(net)
    space.release_grads()
    net(torch.randn(2, 6)).sum().backward()
    space.gather_grads()
    with torch.no_grad():
        space.flat_param.add_(1.0)
    for p, v in zip(space.params, space.view_like_params(space.flat_param), strict=True):
        assert p.data_ptr() == v.data_ptr()
        assert torch.equal(p.detach(), v)
