"""Node-level kNN gather ops -- mirror of the reference's models/operations.py:19-54.

Same two functions and argument meaning; the gather itself is the gfx950 kernel behind
``sonet_knn_gather_f32`` (one thread per output element, no B x C x M*K int64 index expansion).
Differentiable w.r.t. the feature tensor (backward = a gather over inverse neighbour lists, ``sonet_knn_gather_bwd_*``; training only).
"""
import torch

from sonet_hip import ops as _ops


class _KnnGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, knn_I):
        ctx.save_for_backward(knn_I)
        ctx.M = x.shape[2]
        return _ops.knn_gather(x.contiguous(), knn_I.contiguous())

    @staticmethod
    def backward(ctx, g):
        (knn_I,) = ctx.saved_tensors
        B, C, M, K = g.shape
        if g.is_cuda and g.dtype in (torch.float32, torch.bfloat16) and M == ctx.M and M <= 1024 and M * K <= 14336:
            # gather over per-cloud inverse neighbour lists: fixed summation order, no index expansion, no fill (the kernel's limits: one
            # thread per node, the cloud's M * K indices in 56 KiB of LDS; anything larger takes scatter_add below)
            return _ops.knn_gather_bwd(g.contiguous(), knn_I.contiguous(), ctx.M).to(g.dtype), None
        idx = knn_I.reshape(B, 1, M * K).expand(B, C, M * K)
        gx = torch.zeros((B, C, ctx.M), dtype=g.dtype, device=g.device)
        gx.scatter_add_(2, idx, g.reshape(B, C, M * K))
        return gx, None


def knn_gather_by_indexing(som_node, som_node_knn_I):
    """som_node B x C x M, som_node_knn_I B x M x K (int64) -> B x C x M x K."""
    if som_node.requires_grad and torch.is_grad_enabled():
        return _KnnGather.apply(som_node, som_node_knn_I)
    return _ops.knn_gather(som_node.contiguous(), som_node_knn_I.contiguous())


def knn_gather_wrapper(som_node, som_node_knn_I):
    """Coordinates only (C in {2, 3}), as the reference asserts (operations.py:31)."""
    assert som_node.size()[1] in (2, 3)
    return knn_gather_by_indexing(som_node, som_node_knn_I)
