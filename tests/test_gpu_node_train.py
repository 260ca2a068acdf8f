"""Node-level pieces of the training step on hand-written kernels (so-net_amd/csrc/node_train.hip) against the aten expressions the
reference evaluates: torch.max over the neighbours / nodes with its single-arg-max backward (models/layers.py:350-365,
models/networks.py:197) and the backward of the neighbour gather (models/operations.py:38-54)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 40, 64, 9), (2, 128, 64), (1, 5, 7, 1), (2, 3, 1000, 9)])
def test_lastdim_max_autograd_equals_torch_max(shape, dtype):
    from sonet_hip import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(DEV).to(dtype)
    x[..., 0] = x[..., -1]                                   # ties (bf16 makes more of them by itself): the FIRST maximum takes the gradient
    x1 = x.clone().requires_grad_(True)
    x2 = x.clone().requires_grad_(True)
    y1 = ops.lastdim_max_autograd(x1)
    y2, _ = torch.max(x2, dim=-1)
    assert torch.equal(y1, y2)
    gy = torch.randn(y2.shape, generator=g).to(DEV).to(dtype)
    y1.backward(gy)
    y2.backward(gy)
    # torch.max does not promise WHICH maximum it reports among equals on every backend: compare where the maximum is unique,
    # and require ours to sit on the first maximum everywhere
    first = (x == y2.unsqueeze(-1)).int().argmax(dim=-1, keepdim=True)
    expect = torch.zeros_like(x).scatter_(-1, first, gy.unsqueeze(-1))
    assert torch.equal(x1.grad, expect)
    unique = (x == y2.unsqueeze(-1)).sum(dim=-1, keepdim=True) == 1
    assert torch.equal(torch.where(unique, x1.grad, torch.zeros_like(x)), torch.where(unique, x2.grad, torch.zeros_like(x)))


def test_lastdim_max_autograd_nan_wins_like_torch():
    from sonet_hip import ops
    x = torch.tensor([[1.0, float("nan"), 3.0], [2.0, 5.0, 4.0]], device=DEV)
    y = ops.lastdim_max_autograd(x.clone().requires_grad_(True))
    assert torch.isnan(y[0]) and float(y[1]) == 5.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,M,K", [(4, 384, 64, 9), (2, 3, 64, 9), (3, 17, 30, 5), (1, 8, 1000, 4), (2, 5, 1024, 3), (1, 4, 1024, 14)])
def test_knn_gather_backward_equals_scatter_add(B, C, M, K, dtype):
    from models import operations
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C + M + K)
    x = torch.randn(B, C, M, generator=g).to(DEV).requires_grad_(True)
    I = torch.randint(0, M, (B, M, K), generator=g).to(DEV)
    I[:, :, 0] = torch.arange(M, device=DEV)                  # every node is its own first neighbour, as in the loaders' tables
    y = operations.knn_gather_by_indexing(x, I)
    gy = torch.randn(B, C, M, K, generator=g).to(DEV).to(dtype)
    got = ops.knn_gather_bwd(gy, I, M)
    ref = torch.zeros(B, C, M, dtype=torch.float64, device=DEV)
    ref.scatter_add_(2, I.reshape(B, 1, M * K).expand(B, C, M * K), gy.double().reshape(B, C, M * K))
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(got, ops.knn_gather_bwd(gy, I, M))     # fixed summation order: bitwise reproducible
    if dtype == torch.float32:
        y.backward(gy)
        assert torch.equal(x.grad, got)


def test_knn_gather_backward_beyond_the_kernel_limits_takes_scatter_add():
    """M * K above the inverse-list kernel's LDS budget (56 KiB of indices): the autograd node falls back to scatter_add instead of raising."""
    from models import operations
    B, C, M, K = 1, 3, 512, 32
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, C, M, generator=g).to(DEV).requires_grad_(True)
    I = torch.randint(0, M, (B, M, K), generator=g).to(DEV)
    y = operations.knn_gather_by_indexing(x, I)
    gy = torch.randn(B, C, M, K, generator=g).to(DEV)
    y.backward(gy)
    ref = torch.zeros(B, C, M, dtype=torch.float64, device=DEV)
    ref.scatter_add_(2, I.reshape(B, 1, M * K).expand(B, C, M * K), gy.double().reshape(B, C, M * K))
    assert float((x.grad.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
