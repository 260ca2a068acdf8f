"""Randomised differential checks of the CPU oracle's C restatement (oracle/sonet_oracle.c), beyond the fixed
golden fixtures:

* ``index_max`` against the reference's OWN compiled ``index_max.cpp`` (oracle/_ref/index_max.so, which travels to
  the GPU box) on seeded random shapes with the edge cases the domain has: ties, NaN, +-inf, values <= -1000 (the
  reference's initial running maximum, index_max.cpp:89-91), signed zeros, empty nodes, odd sizes;
* ``som_query_topk`` / ``som_group`` / ``knn_gather`` against the aten expressions the reference itself evaluates
  (util/som.py:245-267, models/networks.py:128-171, models/operations.py:38-54), written out here with
  ``torch.topk(sorted=True)`` -- a legal outcome of the reference's ``sorted=False``.

CPU only; sized to run in seconds."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as O

SHAPES = [(1, 1, 1, 1), (1, 3, 7, 2), (2, 5, 33, 4), (3, 16, 257, 64), (2, 384, 300, 64), (1, 7, 1024, 100), (4, 2, 63, 1)]


def _index_max_case(seed, B, C, N, K, flavour):
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((B, C, N)).astype(np.float32)
    index = rng.integers(0, K, size=(B, N)).astype(np.int32)
    if flavour == "ties":
        data = np.round(data * 2).astype(np.float32) / 2                       # many equal values: lowest n must win
    elif flavour == "special":
        pick = rng.random((B, C, N))
        data[pick < 0.05] = np.nan
        data[(pick >= 0.05) & (pick < 0.10)] = np.inf
        data[(pick >= 0.10) & (pick < 0.15)] = -np.inf
        data[(pick >= 0.15) & (pick < 0.20)] = -1000.0                         # equal to the initial maximum: never wins
        data[(pick >= 0.20) & (pick < 0.25)] = -1000.5
        data[(pick >= 0.25) & (pick < 0.30)] = 0.0
        data[(pick >= 0.30) & (pick < 0.35)] = -0.0
    elif flavour == "empty":
        index[:] = index % max(1, K // 2)                                       # upper half of the nodes stays empty
    elif flavour == "low":
        data -= 2000.0                                                          # nothing beats -1000: every output 0
    return data, index


@pytest.mark.parametrize("flavour", ["plain", "ties", "special", "empty", "low"])
@pytest.mark.parametrize("shape", SHAPES)
def test_index_max_restatement_vs_compiled_reference(shape, flavour):
    if O.ref_module() is None:
        pytest.skip("oracle/_ref/index_max.so not built")
    B, C, N, K = shape
    data, index = _index_max_case(1000 * SHAPES.index(shape) + len(flavour), B, C, N, K, flavour)
    ref = O.ref_index_max(data, index, K)
    np.testing.assert_array_equal(O.index_max(data, index, K), ref)
    np.testing.assert_array_equal(O.ref_index_max(data, index, K, threads=2), ref)   # index_max.cpp:51-67 agrees with :73-112
    if flavour == "low":
        assert not ref.any()


def _aten_query_topk(x, node, k):
    """util/som.py:245-267 on aten CPU with sorted=True; k-major min_idx, per-node counts and occupancy."""
    B, _, N = x.shape
    M = node.shape[2]
    diff = x.unsqueeze(3) - node.unsqueeze(2)                                   # B x 3 x N x M
    diff_norm = (diff ** 2).sum(dim=1)
    _, min_idx = torch.topk(diff_norm, k=k, dim=2, largest=False, sorted=True)  # B x N x k
    min_idx = min_idx.permute(0, 2, 1).reshape(B, k * N)
    mask = torch.nn.functional.one_hot(min_idx, M).int()                         # B x kN x M
    return min_idx, mask.sum(1), mask.max(1)[0], mask


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("B,N,M", [(1, 1, 4), (2, 100, 16), (3, 333, 64), (2, 1024, 64), (1, 50, 121)])
def test_som_assign_and_group_restatement_vs_aten(B, N, M, k):
    if k > M:
        pytest.skip("k > M")
    g = torch.Generator().manual_seed(B * 1000 + N + k)
    x = torch.rand(B, 3, N, generator=g) * 2 - 1
    node = torch.rand(B, 3, M, generator=g) * 2 - 1
    min_idx_t, count_t, row_max_t, mask = _aten_query_topk(x, node, k)
    min_idx, count, row_max = O.som_query_topk(x.numpy(), node.numpy(), k)
    np.testing.assert_array_equal(min_idx, min_idx_t.numpy())
    np.testing.assert_array_equal(count, count_t.numpy())
    np.testing.assert_array_equal(row_max, row_max_t.numpy())
    np.testing.assert_array_equal(O.mask_from_min_idx(min_idx, M), mask.numpy())

    # grouping block, models/networks.py:128-171, as the reference writes it (dense mask arithmetic)
    x_stack = x.repeat(1, 1, k)                                                  # :131-135  B x 3 x kN
    mask_f = mask.float().unsqueeze(1)                                           # B x 1 x kN x M
    cnt = mask_f.sum(2)                                                          # B x 1 x M
    som_node = (x_stack.unsqueeze(3) * mask_f).sum(2) / (cnt + 1e-5)             # :141-142
    centers = (som_node.unsqueeze(2) * mask_f).sum(3)                            # :168-169
    som_node_o, centers_o, x_dec_o = O.som_group(x.numpy(), min_idx, M, k)
    # the mean is a float sum in a different order: 1e-5 of the coordinate scale; gather and subtraction are exact on it
    np.testing.assert_allclose(som_node_o, som_node.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_array_equal(centers_o, np.take_along_axis(som_node_o, np.broadcast_to(min_idx[:, None, :], (B, 3, k * N)), axis=2))
    np.testing.assert_array_equal(x_dec_o, x_stack.numpy() - centers_o)
    np.testing.assert_allclose(centers_o, centers.numpy(), rtol=0, atol=1e-5)


def test_som_assign_restatement_ties_go_to_the_lowest_node_id():
    """Duplicate nodes produce exact distance ties; aten's sorted topk and the restatement agree on the node set, and
    the restatement's canonical order (distance, then id) is checked directly."""
    x = torch.tensor([[[0.0, 1.0], [0.0, 0.0], [0.0, 0.0]]])                      # two points
    node = torch.tensor([[[1.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]]])   # nodes 0 == 1, 2 == 3
    min_idx, count, row_max = O.som_query_topk(x.numpy(), node.numpy(), 2)
    np.testing.assert_array_equal(min_idx.reshape(1, 2, 2), [[[2, 0], [3, 1]]])   # k-major: slot 0 of both points, then slot 1
    np.testing.assert_array_equal(count, [[1, 1, 1, 1]])
    np.testing.assert_array_equal(row_max, [[1, 1, 1, 1]])


@pytest.mark.parametrize("B,C,M,K", [(1, 1, 1, 1), (2, 3, 64, 9), (2, 384, 64, 9), (3, 5, 17, 4)])
def test_knn_gather_restatement_vs_aten(B, C, M, K):
    g = torch.Generator().manual_seed(C * 7 + K)
    x = torch.randn(B, C, M, generator=g)
    I = torch.randint(0, M, (B, M, K), generator=g)
    ref = torch.gather(x, 2, I.reshape(B, 1, M * K).expand(B, C, M * K)).reshape(B, C, M, K)   # operations.py:45-54
    np.testing.assert_array_equal(O.knn_gather(x.numpy(), I.numpy()), ref.numpy())
