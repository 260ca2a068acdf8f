"""Synthetic ModelNet40-shaped workload: clouds, SOM nodes, node kNN table and seeded weights.

There is no dataset on the build or GPU boxes, so the benchmark, the smoke test and the parity
fixtures all draw their inputs here:

* ``pc``  ~ U(-1, 1)            B x 3 x N  f32  (ModelNet clouds are unit-sphere normalised)
* ``sn``  = normalised N(0, 1)  B x 3 x N  f32  (surface normals)
* ``node``                      B x 3 x M  f32  -- ``'uniform'``: U(-1,1) (stress: leaves empty
  nodes); ``'som'``: M jittered points of the cloud (what a trained per-shape SOM looks like)
* ``node_knn_I``                B x M x K' i64  -- exact self-kNN of the nodes, ascending distance,
  what the reference loaders compute with faiss (data/modelnet_shrec_loader.py:257-259)

Weights: ``fill_state_dict_`` overwrites every tensor of a reference-keyed state_dict from a
per-key seed, so the reference model (oracle side) and the MI355X model (product side) can be
given identical, non-trivial parameters (BN statistics included) without shipping weight files.
"""
import math
import zlib

import torch


def make_inputs(B, N, M=64, som_k=9, seed=0, node_kind="som", device="cpu"):
    g = torch.Generator().manual_seed(int(seed))
    pc = torch.rand(B, 3, N, generator=g) * 2 - 1
    sn = torch.randn(B, 3, N, generator=g)
    sn = sn / sn.norm(dim=1, keepdim=True).clamp_min(1e-12)
    if node_kind == "uniform":
        node = torch.rand(B, 3, M, generator=g) * 2 - 1
    elif node_kind == "som":
        pick = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)])      # B x M
        node = torch.gather(pc, 2, pick.unsqueeze(1).expand(B, 3, M))
        node = node + 0.05 * torch.randn(B, 3, M, generator=g)
    else:
        raise ValueError(node_kind)
    d = ((node.unsqueeze(3) - node.unsqueeze(2)) ** 2).sum(dim=1)                         # B x M x M
    _, knn_I = torch.topk(d, k=som_k, dim=2, largest=False, sorted=True)
    label = torch.randint(0, 40, (B,), generator=g)
    out = dict(pc=pc.contiguous(), sn=sn.contiguous(), node=node.contiguous(),
               node_knn_I=knn_I.contiguous(), label=label)
    return {k: v.to(device) for k, v in out.items()}


def fill_state_dict_(sd, seed=0):
    """Deterministically overwrite every tensor of ``sd`` (reference key names) in place."""
    for key, t in sd.items():
        g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
        if key.endswith("num_batches_tracked"):
            t.zero_()
            continue
        shape = tuple(t.shape)
        if key.endswith("running_mean"):
            v = 0.2 * torch.randn(shape, generator=g)
        elif key.endswith("running_var"):
            v = 0.5 + torch.rand(shape, generator=g)
        elif ".norm." in key and key.endswith("weight"):
            v = 0.5 + torch.rand(shape, generator=g)
        elif ".norm." in key and key.endswith("bias"):
            v = 0.4 * torch.rand(shape, generator=g) - 0.2
        elif key.endswith("weight"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)      # layers.py:271-280
        elif key.endswith("bias"):
            v = 0.2 * torch.rand(shape, generator=g) - 0.1
        else:
            v = torch.randn(shape, generator=g)
        t.copy_(v.to(t.dtype))
    return sd
