"""Round-5 training kernels against DIRECT references: float64 restatements of the reference math and the C oracle's
``index_max`` -- not older HIP kernels.  tests/test_gpu_segpool.py pins the same kernels bit for bit to the kernels they replaced; here the
link to the reference is one step long:

  sonet_pointmlp_h3_segpool_f32      models/layers.py:431 (Conv1d 1x1 + bias) + models/networks.py:180-185 (index_max + masked gather)
  sonet_pointmlp_h3_stats_xaff_f32   models/layers.py:282-296 on act(BN(raw)) inputs, batch statistics of the output (models/layers.py:60-70)
  sonet_pointmlp_x3_bnb_f32          backward of F.batch_norm(training=True) + ReLU (what autograd runs for models/layers.py:286-294)
                                     followed by the Conv1d input gradient
  sonet_pointmlp_x3_bnb_acc_f32      ... + the other consumer's gradient (autograd's accumulation at models/layers.py:431's concat)

Floats: |got - ref| <= tol * max(|ref|, rms(ref)), tol 1e-5 forward / 1e-4 backward.  Positions: equal to ``oracle_index_max_f32`` (the C
restatement of models/index_max_ext/index_max.cpp:97-109) run on the float64 output rounded to float32 wherever the winner's lead over the
runner-up exceeds the arithmetic's error; everywhere, the chosen column's float64 value is the segment's maximum to within that error."""
import numpy as np
import pytest
import torch

from conftest import assert_close_rms

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _sorted_ids(B, L, M, gen):
    ids = torch.randint(0, M, (B, L), generator=gen, dtype=torch.int32)
    if M > 3:
        ids[ids == 2] = 3                                          # node 2 is empty
    return torch.sort(ids, dim=1).values.contiguous()


@pytest.mark.parametrize("B,C1,C2,Cout,L,M", [(8, 64, 256, 384, 15000, 64), (3, 64, 0, 128, 1001, 7)])
def test_sorted_pool_layer_vs_float64_and_the_oracle_index_max(B, C1, C2, Cout, L, M):
    from oracle import cpu_oracle as O
    from sonet_hip import ops
    gen = torch.Generator().manual_seed(B + L + M)
    x1 = torch.randn(B, C1, L, generator=gen)
    x2 = torch.randn(B, C2, L, generator=gen) if C2 else None
    W = torch.randn(Cout, C1 + C2, generator=gen) * (C1 + C2) ** -0.5
    bias = torch.randn(Cout, generator=gen) * 0.1
    ids = _sorted_ids(B, L, M, gen)
    pos0 = torch.randint(0, L, (B,), generator=gen, dtype=torch.int32)
    row_max = torch.stack([(torch.bincount(ids[b].long(), minlength=M) > 0) for b in range(B)]).to(torch.int32)   # util/som.py:267
    with ops.precision("h3"):
        wp = ops.pointmlp_pack(W.to(DEV), "h3")
        assert ops.pointmlp_h3_segpool_ok(x1.to(DEV), None if x2 is None else x2.to(DEV), wp, Cout, M)
        idx, val = ops.pointmlp_h3_segpool(x1.to(DEV), wp, ops.const_vec(Cout, 1.0, DEV), bias.to(DEV), False, Cout, ids.to(DEV), pos0.to(DEV), M,
                                           row_max.to(DEV), x2=None if x2 is None else x2.to(DEV))
    idx, val = idx.cpu().long(), val.cpu().double()
    # float64 layer (GPU torch, plain matmul)
    xc = (x1 if x2 is None else torch.cat((x1, x2), dim=1)).to(DEV).double()
    y64 = (torch.einsum("oc,bcl->bol", W.to(DEV).double(), xc) + bias.to(DEV).double().view(1, -1, 1)).cpu()
    rms = float(y64.pow(2).mean().sqrt())
    tol = 1e-5 * rms
    # values: the layer's output at the chosen column
    assert_close_rms(val.numpy(), y64.gather(2, idx).numpy(), 1e-5, "pooled values vs the float64 layer at the chosen columns")
    # positions vs the oracle's index_max on the float64 output (as float32): "position 0" of the reference is pos0[b] here
    ref_idx = torch.from_numpy(O.index_max(y64.float().numpy(), ids.numpy(), M)).long()
    occupied = row_max.bool().unsqueeze(1).expand(B, Cout, M)
    seg_max = torch.full((B, Cout, M), -float("inf"), dtype=torch.float64)
    seg_max.scatter_reduce_(2, ids.long().unsqueeze(1).expand(B, Cout, L), y64, "amax", include_self=True)
    beaten = occupied & (seg_max > -1000.0)
    # (a) where something beat -1000: the chosen column belongs to the node and holds the node's maximum up to the arithmetic's error
    chosen_node = ids.long().unsqueeze(1).expand(B, Cout, L).gather(2, idx)
    node_of_bin = torch.arange(M).view(1, 1, M).expand(B, Cout, M)
    assert bool((chosen_node == node_of_bin)[beaten].all())
    assert bool((y64.gather(2, idx)[beaten] >= seg_max[beaten] - 2 * tol).all())
    # (b) bit-exact positions against the oracle wherever the lead is clear (runner-up more than 4 tol behind)
    y_wo = y64.clone()
    y_wo.scatter_(2, ref_idx, -float("inf"))
    second = torch.full((B, Cout, M), -float("inf"), dtype=torch.float64)
    second.scatter_reduce_(2, ids.long().unsqueeze(1).expand(B, Cout, L), y_wo, "amax", include_self=True)
    clear = beaten & (seg_max - second > 4 * tol)
    assert float(clear.float().mean()) > 0.9 * float(beaten.float().mean())
    assert torch.equal(idx[clear], ref_idx[clear])
    # (c) bins nothing beat / empty nodes: models/networks.py:185 gathers original column 0 = sorted position pos0[b]
    assert torch.equal(idx[~beaten], pos0.long().view(B, 1, 1).expand(B, Cout, M)[~beaten])


@pytest.mark.parametrize("B,C1,C2,Cout,L", [(8, 64, 0, 128, 15000), (4, 64, 256, 384, 4100)])
def test_layer_with_normalise_on_load_vs_float64(B, C1, C2, Cout, L):
    from sonet_hip import ops
    gen = torch.Generator().manual_seed(C1 + C2 + L)

    def raw_case(C):
        return (torch.randn(B, C, L, generator=gen) * 1.5).to(DEV), (torch.rand(C, generator=gen) + 0.5).to(DEV), (torch.randn(C, generator=gen) * 0.4).to(DEV)
    r1, s1, h1 = raw_case(C1)
    r2, s2, h2 = raw_case(C2) if C2 else (None, None, None)
    W = (torch.randn(Cout, C1 + C2, generator=gen) * (C1 + C2) ** -0.5).to(DEV)
    bias = (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
    with ops.precision("h3"):
        wp = ops.pointmlp_pack(W, "h3")
        xa = (s1, h1, True) + ((s2, h2, True) if C2 else ())
        y, mean, var = ops.pointmlp_stats(r1, wp, ops.const_vec(Cout, 1.0, DEV), bias, False, Cout, x2=r2, xaff=xa)
    x64 = torch.relu(r1.double() * s1.double().view(1, -1, 1) + h1.double().view(1, -1, 1))
    if C2:
        x64 = torch.cat((x64, torch.relu(r2.double() * s2.double().view(1, -1, 1) + h2.double().view(1, -1, 1))), dim=1)
    y64 = torch.einsum("oc,bcl->bol", W.double(), x64) + bias.double().view(1, -1, 1)
    assert_close_rms(y.cpu().numpy(), y64.cpu().numpy(), 1e-5, "layer output")
    assert_close_rms(mean.cpu().numpy(), y64.mean(dim=(0, 2)).cpu().numpy(), 1e-5, "batch mean")
    assert_close_rms(var.cpu().numpy(), y64.var(dim=(0, 2), unbiased=False).cpu().numpy(), 1e-5, "biased batch variance")


def _bn_relu_backward_case(B, C, Cout, L, relu, seed):
    """A BatchNorm(+ReLU) layer in training mode and the 1x1 convolution BELOW it.  raw = the layer's pre-norm output, gy = gradient of
    its activation; what the launch under test must produce: g_raw = dL/d raw (F.batch_norm backward, batch statistics) and W^T g_raw."""
    gen = torch.Generator().manual_seed(seed)
    raw = (torch.randn(B, C, L, generator=gen) * 1.5 + torch.randn(1, C, 1, generator=gen)).to(DEV)
    gy = (torch.randn(B, C, L, generator=gen) * 1e-3).to(DEV)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(DEV), (torch.randn(C, generator=gen) * 0.3).to(DEV)
    Wl = (torch.randn(C, Cout, generator=gen) * Cout ** -0.5).to(DEV)           # the layer's weight: C outputs from Cout inputs
    raw64 = raw.double().requires_grad_(True)
    act = torch.nn.functional.batch_norm(raw64, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    act = torch.relu(act) if relu else act
    (g_raw64,) = torch.autograd.grad(act, raw64, gy.double())
    gx64 = torch.einsum("co,bcl->bol", Wl.double(), g_raw64)
    return raw, gy, gamma, beta, Wl, g_raw64, gx64


def _bnb_coefficients(ops, raw, gy, gamma, beta, relu, eps=1e-5):
    """The per-channel coefficients the product computes in its own statistics pass + finalize (forward scale / shift, backward a, b, c0)."""
    mean, var = ops.channel_stats(raw)
    invstd, sc, sh = ops.bn_fwd_coeffs(mean, var, gamma, beta, eps)
    sums = ops.pointwise_bwd_stats(gy, raw, sc, sh, relu, want_sums=True)
    a, b, c0 = ops.bn_bwd_coeffs(sums, mean, invstd, gamma, raw.shape[0] * raw.shape[2])[:3]
    return sc, sh, a, b, c0


@pytest.mark.parametrize("B,C,Cout,L,relu", [(8, 256, 128, 15000, True), (3, 128, 64, 577, False)])
def test_dgrad_with_batchnorm_backward_on_load_vs_float64_autograd(B, C, Cout, L, relu):
    from sonet_hip import ops
    raw, gy, gamma, beta, Wl, g_raw64, gx64 = _bn_relu_backward_case(B, C, Cout, L, relu, B + C + L)
    sc, sh, a, b, c0 = _bnb_coefficients(ops, raw, gy, gamma, beta, relu)
    wpt = ops.pointmlp_pack(Wl.t().contiguous(), "x3")
    one, zero = ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV)
    gx, g_raw = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout)
    assert_close_rms(g_raw.cpu().numpy(), g_raw64.cpu().numpy(), 1e-4, "g_raw vs float64 autograd of batch_norm + relu")
    assert_close_rms(gx.cpu().numpy(), gx64.cpu().numpy(), 1e-4, "input gradient vs float64")


@pytest.mark.parametrize("B,C,Cout,L,relu", [(8, 128, 64, 15000, True)])
def test_dgrad_with_accumulating_store_vs_float64_autograd(B, C, Cout, L, relu):
    from sonet_hip import ops
    raw, gy, gamma, beta, Wl, g_raw64, gx64 = _bn_relu_backward_case(B, C, Cout, L, relu, B + C + L + 1)
    other = (torch.randn(B, Cout, L, generator=torch.Generator().manual_seed(5)) * 1e-3).to(DEV)
    other[:, :, ::7] = 0.0
    sc, sh, a, b, c0 = _bnb_coefficients(ops, raw, gy, gamma, beta, relu)
    wpt = ops.pointmlp_pack(Wl.t().contiguous(), "x3")
    one, zero = ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV)
    gx, g_raw = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout, acc=other)
    assert_close_rms(g_raw.cpu().numpy(), g_raw64.cpu().numpy(), 1e-4, "g_raw")
    assert_close_rms(gx.cpu().numpy(), (gx64 + other.double()).cpu().numpy(), 1e-4, "input gradient + the other consumer's gradient")
