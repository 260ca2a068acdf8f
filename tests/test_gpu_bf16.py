"""GPU tests of the bf16 path (BASELINE configs[1] "bf16"): bf16-storage point-wise layers on bf16 MFMA, index_max on bf16
features, the encoder forward in bf16.

The reference is f32-only (models/index_max_ext/index_max_cuda.cu:75-76 dispatches float), so the bar is the one
SURVEY.md 7 (step 5) sets: integer outputs bit-exact ON THE SAME bf16 INPUTS, float outputs within a stated bf16 tolerance of
the f32 fixture:  |got - ref| <= BF16_TOL * max(|ref|, rms(ref))  with BF16_TOL = 5e-2 for the WORST element (bf16 has 8
significand bits, 2^-9 = 2e-3 per rounding; the encoder rounds seven layer outputs in a row and passes three max-pools;
measured worst elements on the three fixtures: 1.4e-2 .. 3.3e-2, printed by the test)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import assert_close_rms, golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_TOL = 5e-2


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def ref_layer(x1, x2, W, scale, shift, relu, gidx=None):
    """The layer on the SAME bf16 operands, float64 accumulate (what the MFMA computes up to summation order)."""
    xs = x1.double()
    if gidx is not None:
        B, C1, L1 = x1.shape
        idx = gidx.long()
        ok = (idx >= 0) & (idx < L1)
        xs = torch.gather(xs, 2, idx.clamp(0, L1 - 1).unsqueeze(1).expand(B, C1, idx.shape[1])) * ok.unsqueeze(1)
    if x2 is not None:
        xs = torch.cat((xs, x2.double()), dim=1)
    Wb = W.to(torch.bfloat16).double()
    y = torch.einsum("oc,bcl->bol", Wb, xs) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("B,C1,C2,Cout,L,relu", [
    (2, 6, 0, 64, 300, True), (3, 64, 0, 128, 1000, True), (2, 64, 256, 384, 1501, False), (2, 64, 256, 384, 1500, False),
    (1, 387, 0, 512, 576, True), (2, 512, 0, 512, 64, True), (2, 515, 0, 768, 64, True), (1, 768, 0, 1024, 64, False),
    (2, 16, 3, 32, 1, False), (1, 128, 0, 256, 15000, True), (2, 48, 0, 96, 130, True)])
def test_pointmlp_bf16_vs_same_operand_reference(B, C1, C2, Cout, L, relu):
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 * 7 + L)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16) if C2 else None
    if C2 and C1 % 16:
        pytest.skip("second panel needs C1 % 16 == 0")
    W = torch.randn(Cout, C1 + C2, generator=g) * (2.0 / (C1 + C2)) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    wp = ops.pointmlp_pack(W.to(DEV), "bf16")
    y = ops.pointmlp(x1.to(DEV), wp, scale.to(DEV), shift.to(DEV), relu, Cout, x2=x2.to(DEV) if x2 is not None else None)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (B, Cout, L)
    ref = ref_layer(x1, x2, W, scale, shift, relu)
    got = y.cpu().double()
    # the f32 accumulator differs from the float64 one by summation order only: the bf16 rounding of the result may differ by
    # one unit in the last place (2^-8 relative) where the exact value sits near a rounding boundary
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-6 * ref.abs().max()
    assert bool((err <= bound).all()), "worst err/bound %.3g" % float((err / bound).max())
    # and on average it is the correctly rounded value
    exact = ref.to(torch.bfloat16).double()
    assert float((got != exact).double().mean()) < 0.02


def test_pointmlp_bf16_gather_matches_materialised():
    from sonet_hip import ops
    g = torch.Generator().manual_seed(9)
    B, C1, M, K = 2, 384, 64, 9
    x1 = torch.randn(B, C1, M, generator=g).to(torch.bfloat16)
    gidx = torch.randint(-1, M + 1, (B, K * M), generator=g, dtype=torch.int32)         # -1 and M: out of range -> zeros
    x2 = torch.randn(B, 3, K * M, generator=g).to(torch.bfloat16)
    W = torch.randn(512, C1 + 3, generator=g) * 0.07
    scale, shift = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g) * 0.3
    wp = ops.pointmlp_pack(W.to(DEV), "bf16")
    y = ops.pointmlp(x1.to(DEV), wp, scale.to(DEV), shift.to(DEV), True, 512, x2=x2.to(DEV), gidx=gidx.to(DEV))
    ref = ref_layer(x1, x2, W, scale, shift, True, gidx=gidx)
    err = (y.cpu().double() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-6 * ref.abs().max()
    assert bool((err <= bound).all())


def test_index_max_gather_bf16_same_input_bit_exact():
    """bf16 features make exact ties common: positions must equal the reference scan on the same (widened) values."""
    from oracle import cpu_oracle as O
    from sonet_hip import ops
    g = torch.Generator().manual_seed(4)
    B, C, N, K = 3, 384, 15000, 64
    data = (torch.randn(B, C, N, generator=g) * 0.5).to(torch.bfloat16)
    index = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32)
    index[1, :] = index[1, :] % 7                                                       # leaves empty nodes
    row_max = torch.stack([(torch.bincount(index[b].long(), minlength=K) > 0) for b in range(B)]).to(torch.int32)
    ref = O.index_max(data.float().numpy(), index.numpy(), K)
    idx, val = ops.index_max_gather(data.to(DEV), index.to(DEV), K, row_max.to(DEV))
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    exp = torch.gather(data.float(), 2, torch.from_numpy(ref).long() * row_max.unsqueeze(1).long())
    np.testing.assert_array_equal(val.cpu().numpy(), exp.numpy())
    np.testing.assert_array_equal(ops.index_max(data.to(DEV), index.to(DEV), K).cpu().numpy(), ref)


@pytest.mark.parametrize("B,Cin0,L", [(2, 6, 300), (3, 6, 15000), (1, 6, 1), (2, 3, 129), (1, 6, 257), (5, 6, 1024)])
def test_pointresnet_bf16_fused_vs_layerwise(B, Cin0, L):
    """One-kernel bf16 first PointNet == the four bf16 layer launches up to one bf16 unit in the last place on a few
    elements (same operands and roundings; the chained channel order changes the summation order inside a K chunk)."""
    from models import layers as Lm
    from sonet_hip import ops, synth
    pr = Lm.PointResNet(Cin0, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    x = torch.randn(B, Cin0, L, generator=torch.Generator().manual_seed(L)).to(DEV)
    old = ops.FUSE_POINTRESNET
    try:
        with ops.precision("bf16"), torch.no_grad():
            ops.FUSE_POINTRESNET = True
            with ops.kernel_timing() as rec:
                y_fused = pr(x)
            assert any(n.startswith("pointresnet_bf16") for n, _, _ in rec.records), "fused bf16 kernel did not run"
            ops.FUSE_POINTRESNET = False
            y_layer = pr(x)
    finally:
        ops.FUSE_POINTRESNET = old
    assert y_fused.dtype == torch.bfloat16 and y_layer.dtype == torch.bfloat16 and tuple(y_fused.shape) == (B, 384, L)
    a, b = y_fused.float().cpu().double(), y_layer.float().cpu().double()
    rms = float(b.pow(2).mean().sqrt())
    err = (a - b).abs()
    # a flipped bf16 rounding in a hidden layer moves an output by a few parts in a thousand of the output scale
    assert bool((err <= 2.0 ** -7 * b.abs() + 2.0 ** -7 * rms).all()), "worst %.3g of rms" % float((err / rms).max())
    assert float((a != b).double().mean()) < 0.25


@pytest.mark.parametrize("B,N,M,k,kind", [(3, 5000, 64, 3, "som"), (2, 333, 64, 3, "uniform"), (2, 40, 64, 1, "uniform"), (1, 1, 64, 3, "uniform"),
                                          (4, 700, 16, 2, "som"), (2, 2000, 100, 3, "uniform")])
def test_pointresnet_bf16_pool_equals_store_plus_index_max(B, N, M, k, kind):
    """Pool variant (nothing of first_pn_out reaches HBM) == index_max_gather_bf16 of what the store variant writes, bit for bit
    (same MFMA order, bf16 rounding is monotone), including empty nodes (features of original copy 0)."""
    from models import layers as Lm
    from sonet_hip import ops, synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    inp = synth.make_inputs(B, N, M=M, som_k=min(9, M), seed=N + k, node_kind=kind)
    x, sn = inp["pc"].to(DEV), inp["sn"].to(DEV)
    a = ops.som_assign(x, inp["node"].to(DEV), k)
    sg = ops.som_sort_group(x, sn, a)
    with ops.precision("bf16"), torch.no_grad():
        wstream, affine = pr._fused_state()
        first_sorted = ops.pointresnet_bf16(sg["x_aug_sorted"], wstream, affine)         # bf16, node-sorted columns
        got = ops.pointresnet_bf16_pool(sg, wstream, affine, M)
    # reference semantics on the sorted tensor: positions are in sorted order, an empty node gathers ORIGINAL copy 0 = sorted pos0
    idx, _ = ops.index_max_gather(first_sorted, sg["ids_sorted"], M, None)
    val = torch.gather(first_sorted.float(), 2, idx.long())
    beat = torch.gather(first_sorted.float(), 2, idx.long()) > -1000.0
    occupied = (a.count > 0).unsqueeze(1)
    copy0 = torch.gather(first_sorted.float(), 2, sg["pos0"].long().view(B, 1, 1).expand(B, 384, 1)).expand(B, 384, M)
    exp = torch.where(occupied & beat, val, copy0)
    assert tuple(got.shape) == (B, 384, M)
    assert torch.equal(got, exp), "max |diff| %.3g at %d elements" % (float((got - exp).abs().max()), int((got != exp).sum()))


def make_opt(g, B, N):
    return Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True,
                     feature_num=1024, activation="relu", normalization="batch", dropout=0.7, node_num=64,
                     k=int(g["k"]), som_k=int(g["som_k"]), som_k_type=str(g["som_k_type"]), bn_momentum=0.1,
                     bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)


@pytest.mark.parametrize("case", ["classifier_b2_n5000", "classifier_b8_n1024", "classifier_b2_n256"])
def test_encoder_classifier_forward_bf16(case):
    """configs[1] shape (and two smaller ones) in bf16: node assignment bit-exact vs the fixture (it is computed in f32),
    arg-max positions bit-exact against the reference scan of the SAME bf16 first_pn_out, float outputs within BF16_TOL."""
    from models import networks as NW
    from oracle import cpu_oracle as O
    from sonet_hip import ops, synth
    g = golden(case)
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    opt = make_opt(g, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), seed)
    synth.fill_state_dict_(cls.state_dict(), seed + 1)
    enc.to(DEV).eval()
    cls.to(DEV).eval()
    enc.want_first_pn_out = True                                   # keep first_pn_out (bf16) for the same-input index check
    with ops.precision("bf16"), torch.no_grad(), ops.kernel_timing() as rec:
        feat = enc(cu(g["pc"]), cu(g["sn"]), cu(g["node"]), cu(g["node_knn_I"]), is_train=False)
        score = cls(feat)
    names = [n for n, _, _ in rec.records]
    assert any(n.startswith("pointmlpbf16") for n in names) and not any(n.startswith(("pointmlph3", "pointmlpx3", "pointresnet_fused")) for n in names), names
    assert any(n.startswith("pointresnet_bf16") for n in names), names
    np.testing.assert_array_equal(enc.min_idx.cpu().numpy(), g["min_idx"])
    first = enc.first_pn_out
    assert first.dtype == torch.bfloat16
    ref_pos = O.index_max(first.float().cpu().numpy(), enc.min_idx.int().cpu().numpy(), 64)
    got_pos, got_val = ops.index_max_gather(first, enc.min_idx.int().contiguous(), 64, (enc._lazy["a"].count > 0).to(torch.int32))
    np.testing.assert_array_equal(got_pos.cpu().numpy(), ref_pos)
    np.testing.assert_array_equal(got_val.cpu().numpy(), enc.first_pn_out_masked_max.float().cpu().numpy())
    worst = {}

    def close(name, got, ref):
        got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
        rms = np.sqrt(np.mean(ref ** 2))
        worst[name] = float((np.abs(got - ref) / np.maximum(np.abs(ref), rms)).max())
        assert_close_rms(got, ref, BF16_TOL, name)

    try:
        close("som_node", enc.som_node.cpu().numpy(), g["som_node"])
        close("first_pn_out", first[:, ::16, ::5].float().cpu().numpy(), g["first_pn_out_sub"])
        close("masked_max", enc.first_pn_out_masked_max.float().cpu().numpy(), g["first_pn_out_masked_max"])
        close("knn_feature_1", enc.knn_feature_1[:, ::4].float().cpu().numpy(), g["knn_feature_1"])
        close("final_pn_out", enc.final_pn_out[:, ::4].float().cpu().numpy(), g["final_pn_out"])
        close("feature", feat.float().cpu().numpy(), g["feature"])
        close("score", score.float().cpu().numpy(), g["score"])
    finally:
        print("bf16 worst err / max(|ref|, rms):", {k: "%.2e" % v for k, v in worst.items()})
    # bf16 keeps the decision: same predicted class as the f32 reference wherever the reference's margin is not razor thin
    ref_score = torch.from_numpy(g["score"])
    top2 = ref_score.topk(2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 0.05 * ref_score.abs().max()
    assert bool((score.cpu().argmax(1) == ref_score.argmax(1))[clear].all())


# ------------------------------------------------------------------------------------------ bf16 training (configs[1]: forward + backward)
def test_bf16_layer_backward_vs_float64():
    """Training-mode layer (bf16 conv -> batch statistics -> normalise + ReLU) forward and backward in bf16 storage against
    float64 autograd of the same math on the same bf16-rounded inputs: bf16 tolerance (2e-2 of the tensor's rms)."""
    from models import layers as Lm
    from sonet_hip import ops, synth
    layer = Lm.EquivariantLayer(64, 128, "relu", "batch", 0.1)
    synth.fill_state_dict_(layer.state_dict(), 5)
    layer.to(DEV).train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(6, 64, 2000, generator=g).to(torch.bfloat16)
    gy = torch.randn(6, 128, 2000, generator=g).to(torch.bfloat16)
    xd = x.to(DEV).requires_grad_(True)
    with ops.precision("bf16"), ops.kernel_timing() as rec:
        y = layer(xd)
        y.backward(gy.to(DEV))
    names = [n for n, _, _ in rec.records]
    assert y.dtype == torch.bfloat16 and xd.grad.dtype == torch.bfloat16
    assert any(n.startswith("pointmlpbf16") for n in names) and "pointwise_bwd_apply_bf16" in names and "channel_stats_bf16" in names, names
    # float64 reference
    W = layer.conv.weight.detach().cpu().double()[:, :, 0].to(torch.bfloat16).double().requires_grad_(True)
    b64 = layer.conv.bias.detach().cpu().double().requires_grad_(True)
    gam, bet = layer.norm.weight.detach().cpu().double().requires_grad_(True), layer.norm.bias.detach().cpu().double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    raw = torch.einsum("oc,bcl->bol", W, x64) + b64.view(1, -1, 1)
    mean, var = raw.mean(dim=(0, 2), keepdim=True), raw.var(dim=(0, 2), unbiased=False, keepdim=True)
    y64 = torch.relu((raw - mean) / torch.sqrt(var + layer.norm.eps) * gam.view(1, -1, 1) + bet.view(1, -1, 1))
    y64.backward(gy.double())

    def rel(a, r):
        a, r = a.detach().cpu().double(), r.detach().double()
        return float((a - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())

    errs = dict(y=rel(y, y64), gx=rel(xd.grad, x64.grad), gw=rel(layer.conv.weight.grad[:, :, 0], W.grad),
                ggamma=rel(layer.norm.weight.grad, gam.grad), gbeta=rel(layer.norm.bias.grad, bet.grad))
    print("bf16 layer fwd/bwd rel-rms errors:", {k: "%.2e" % v for k, v in errs.items()})
    assert all(v < 2e-2 for v in errs.values()), errs


@pytest.mark.parametrize("mode,tol", [("bf16", 1.5e-2), ("x3", 1e-4)])       # measured: 2e-3 .. 6e-3 and 3e-7 .. 1e-6
def test_bf16_first_pointnet_backward_with_fixed_routing_vs_float64(mode, tol):
    """The whole first PointNet in training mode (four bf16 layers with batch statistics, the skip concat, the last layer +
    per-node arg-max pool as one autograd node with its SPARSE dgrad / wgrad) against float64 autograd of the same network
    gathered at the SAME arg-max positions -- i.e. everything of the bf16 backward except the choice of winners, which a
    reduced-precision forward is free to make differently (see the training-step test below)."""
    from models import layers as Lm
    from sonet_hip import ops, synth
    B, L, M = 4, 1500, 16
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=11)
    pr.to(DEV).train()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 6, L, generator=g).to(torch.bfloat16).float()
    ids = torch.randint(0, M, (B, L), generator=g, dtype=torch.int32)
    wr = torch.randn(B, 384, M, generator=g)
    with ops.precision(mode):
        first, pooled, gi = pr.forward_pooled(x.to(DEV), ids.to(DEV), torch.ones(B, M, dtype=torch.int32, device=DEV), M)
        (pooled * wr.to(DEV)).sum().backward()
    assert first.dtype == (torch.bfloat16 if mode == "bf16" else torch.float32)
    # float64 twin with the same routing
    # The twin rounds where the bf16 path rounds in the FORWARD (weights as MFMA operands, the stored raw and activation
    # tensors; straight-through for the gradient), so that both networks have the same ReLU pattern: sums like d loss / d beta
    # cancel heavily under the sparse pooled gradient and a fraction f of differing masks shows up as ~sqrt(f) relative error
    # (7-11 % against an unrounded float64 forward -- measured -- which says nothing about the backward arithmetic).
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in pr.named_parameters()}
    rnd = (lambda t: t + (t.to(torch.bfloat16).double() - t).detach()) if mode == "bf16" else (lambda t: t)

    def layer(h, i, bn=True):
        raw = rnd(torch.einsum("oc,bcl->bol", rnd(P["layers.%d.conv.weight" % i][:, :, 0]), h) + P["layers.%d.conv.bias" % i].view(1, -1, 1))
        if not bn:
            return raw
        mean, var = raw.mean(dim=(0, 2), keepdim=True), raw.var(dim=(0, 2), unbiased=False, keepdim=True)
        return rnd(torch.relu((raw - mean) / torch.sqrt(var + 1e-5) * P["layers.%d.norm.weight" % i].view(1, -1, 1) + P["layers.%d.norm.bias" % i].view(1, -1, 1)))
    skip = layer(x.double(), 0)
    t = layer(layer(skip, 1), 2)
    y = layer(torch.cat((skip, t), dim=1), 3, bn=False)
    (y.gather(2, gi.cpu().long()) * wr.double()).sum().backward()
    errs = {}
    for k, v in pr.named_parameters():
        r = P[k].grad
        if r is None or float(r.pow(2).mean().sqrt()) < 1e-9:          # a bias in front of a BatchNorm has no gradient
            continue
        a = v.grad.detach().cpu().double()
        errs[k] = float((a - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
    print("%s first PointNet backward, fixed routing, rel-rms vs float64:" % mode, {k: "%.1e" % e for k, e in errs.items()})
    # (the f32-class run of the same harness, x3, agrees to 1e-6)
    assert len(errs) >= 10 and max(errs.values()) < tol, errs


@pytest.mark.parametrize("fixture", ["train_step_b16_n512", "train_step_b8_n5000"])
def test_classifier_training_step_bf16(fixture):
    """One training step of the classifier in bf16 storage (configs[1] shape family) vs the reference's float64 run of the same
    step: loss and feature at bf16 tolerance, gradients aligned with the float64 ones (cosine), running statistics updated."""
    from models import networks as NW
    from sonet_hip import ops, synth
    g = golden(fixture)
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), seed)
    synth.fill_state_dict_(cls.state_dict(), seed + 1)
    enc.to(DEV).train()
    cls.to(DEV).train()
    with ops.precision("bf16"), ops.kernel_timing() as rec:
        feat = enc(cu(g["pc"]), cu(g["sn"]), cu(g["node"]), cu(g["node_knn_I"]), is_train=True, epoch=0)
        score = cls(feat, 0)
        loss = torch.nn.functional.cross_entropy(score, cu(g["label"]))
        loss.backward()
    names = set(n for n, _, _ in rec.records)
    assert any(n.startswith("pointmlpbf16") for n in names) and not any(n.startswith(("pointmlph3", "pointmlpx3")) for n in names), names
    assert {"pointwise_bwd_stats_bf16", "pointwise_bwd_apply_bf16", "channel_stats_bf16", "channel_affine_act_bf16", "pooled_dgrad_mfma", "pooled_wgrad"} <= names, names
    # the pool of the first PointNet's output: the last layer's epilogue (first_pn_out never written), or index_max on the stored tensor
    assert any(n.startswith("pointmlpbf16_pool") for n in names) or "index_max_gather_bf16" in names, names
    # train-mode BatchNorm divides by batch statistics of bf16-rounded activations and the feature passes three max-pools whose
    # winners may change: bound the rms error (4e-2; measured 2.0e-2 at N=512, 3.1e-2 at N=5000) and the worst element
    f_got, f_ref = feat.detach().cpu().double().numpy(), g["feature"].astype(np.float64)
    f_rel = float(np.sqrt(np.mean((f_got - f_ref) ** 2)) / np.sqrt(np.mean(f_ref ** 2)))
    print("bf16 training forward: feature rel-rms error %.2e" % f_rel)
    assert f_rel < 4e-2
    assert_close_rms(f_got, f_ref, 2e-1, "train feature (bf16)")
    assert abs(float(loss.detach()) - float(g["loss"])) <= 3e-2 * max(1.0, abs(float(g["loss"])))
    params = dict(enc.named_parameters())
    # Gradients pass three arg-max pools (per node over its points, over the K' neighbours, over the M nodes).  With 8
    # significand bits, candidates within 2^-8 of the maximum tie or swap, and every swap re-routes that (cloud, channel)'s
    # gradient to another column -- so the end-to-end gradient of EVERY layer (also the ones right behind the loss: measured
    # cosine 0.83-0.88 on all of them, against 1e-3 rel-rms for the f32-class arithmetics) is a different, equally valid
    # sub-gradient, not a noisy copy of the float64 one (0.73-0.81 at N=5000, where a node's pool has ~700 candidates).  Bar here:
    # same direction (cosine above the fixture's measured floor, below).  What pins the bf16 backward arithmetic itself are test_bf16_layer_backward_vs_float64 (1-2e-2
    # per layer) and test_bf16_first_pointnet_backward_with_fixed_routing_vs_float64 (the whole first PointNet incl. the sparse
    # pooled dgrad / wgrad, gathered at the same winners) above.
    cosines = {}
    for k in [k[7:] for k in g.files if k.startswith("grad64/") and not k.startswith("grad64/cls.")]:
        truth = g["grad64/" + k].astype(np.float64)
        if np.sqrt(np.mean(truth ** 2)) < 1e-5 or not k.endswith("conv.weight"):
            continue
        f = params[k].grad.detach().flatten()
        mine = f[::max(1, f.numel() // 16384)].cpu().numpy().astype(np.float64)
        cosines[k] = float(np.dot(mine, truth) / (np.linalg.norm(mine) * np.linalg.norm(truth)))
    print("bf16 training step: gradient cosine vs the float64 reference:", {k: "%.3f" % v for k, v in cosines.items()})
    # gate at what this fixture measures (0.83-0.88 at N = 512, 0.73-0.81 at N = 5000), not at a generic 0.6
    assert min(cosines.values()) > {"train_step_b16_n512": 0.80, "train_step_b8_n5000": 0.70}[fixture], cosines
    assert sum(1 for p in enc.parameters() if p.grad is None) == int(g["dead_grad_count"])
    sd = enc.state_dict()
    for k in [k[3:] for k in g.files if k.startswith("bn/")]:
        assert_close_rms(sd[k].cpu().numpy(), g["bn/" + k], 2e-1, "running stat " + k)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C1,C2,Cout,L", [(3, 64, 0, 128, 1500), (2, 6, 0, 64, 777), (2, 256, 64, 384, 300), (4, 128, 0, 256, 64)])
def test_pointmlp_bf16_statistics_epilogue(B, C1, C2, Cout, L):
    """bf16 layer kernel's statistics epilogue: output bit-identical to the plain launch; mean / biased variance of the STORED bf16
    values equal to the separate pass (f64 sums) to 1e-6 of the channel's scale (odd L: the 2-byte access variant)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = (torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(DEV)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    scale = torch.ones(Cout, device=DEV)
    shift = (torch.randn(Cout, generator=g) * 3.0).to(DEV)
    wp = ops.pointmlp_pack(w, "bf16")
    y0 = ops.pointmlp(x1, wp, scale, shift, False, Cout, x2=x2)
    y1, m1, v1 = ops.pointmlp_stats(x1, wp, scale, shift, False, Cout, x2=x2)
    assert torch.equal(y0, y1)
    ref = y0.double()
    mref, vref = ref.mean(dim=(0, 2)), ref.var(dim=(0, 2), unbiased=False)
    sc = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
    assert float(((m1.double() - mref).abs() / sc).max()) < 1e-6
    assert float(((v1.double() - vref).abs() / sc ** 2).max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,C1,C2,Cout,L,relu", [
    (36, 64, 0, 128, 15000, True), (36, 64, 256, 384, 15000, False), (64, 128, 0, 256, 8192, True), (41, 256, 0, 128, 13002, True),
    (36, 128, 0, 64, 15000, False), (33, 64, 0, 192, 16000, True)])
def test_pointmlp_bf16_streaming_generation(B, C1, C2, Cout, L, relu):
    """Launches of >= 8192 column groups with K a multiple of 64 take the streaming kernel (W slab resident in LDS, persistent waves,
    hand-counted X look-ahead across units): bit-identical to the staged kernel, which the same layer takes on batches of <= 16
    clouds (same K order, same f32 MFMA chain), with and without the statistics epilogue; statistics against f64 sums."""
    from sonet_hip import ops
    assert B * ((L + 63) // 64) >= 8192 and 16 * ((L + 63) // 64) < 8192
    g = torch.Generator().manual_seed(C1 + 3 * Cout + L)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    W = (torch.randn(Cout, C1 + C2, generator=g) * (2.0 / (C1 + C2)) ** 0.5).to(DEV)
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    wp = ops.pointmlp_pack(W, "bf16")
    y = ops.pointmlp(x1, wp, scale, shift, relu, Cout, x2=x2)
    parts = [ops.pointmlp(x1[b0:b0 + 16].contiguous(), wp, scale, shift, relu, Cout, x2=x2[b0:b0 + 16].contiguous() if C2 else None)
             for b0 in range(0, B, 16)]
    ref = torch.cat(parts, dim=0)
    assert torch.equal(y, ref), "streaming and staged kernels differ in %d values" % int((y != ref).sum())
    y2 = ops.pointmlp(x1, wp, scale, shift, relu, Cout, x2=x2)                       # (and run to run)
    assert torch.equal(y, y2)
    ys, m1, v1 = ops.pointmlp_stats(x1, wp, scale, shift, relu, Cout, x2=x2)
    assert torch.equal(ys, y)
    yd = y.double()
    mref, vref = yd.mean(dim=(0, 2)), yd.var(dim=(0, 2), unbiased=False)
    sc = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
    assert float(((m1.double() - mref).abs() / sc).max()) < 1e-6
    assert float(((v1.double() - vref).abs() / sc ** 2).max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["segmenter_b2_n256", "segmenter_b2_n1024"])
def test_segmenter_forward_bf16(case):
    """The part segmenter under precision('bf16') (global switch: SONET_POINTMLP_PRECISION / bench --precision): the node-wise
    layer-1 path receives a mix of f32 (som_node, pooled maxima) and bf16 maps; both data flows run and stay within BF16_TOL of the
    reference fixture, node ids bit-exact."""
    from models import networks as NW
    from sonet_hip import ops, synth
    g = golden(case)
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.6, node_num=64, k=3, som_k=9, som_k_type="center",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=50)
    enc, seg = NW.Encoder(opt), NW.Segmenter(opt)
    synth.fill_state_dict_(enc.state_dict(), seed)
    synth.fill_state_dict_(seg.state_dict(), seed + 1)
    enc.to(DEV).eval()
    seg.to(DEV).eval()
    args = (cu(g["pc"]), cu(g["sn"]), cu(g["label"]), cu(g["node"]), cu(g["node_knn_I"]))
    with ops.precision("bf16"), torch.no_grad():
        seg.nodewise = True
        with ops.kernel_timing() as rec:
            score = NW.segmentation_forward(enc, seg, *args)
        seg.nodewise = False
        score_dense = NW.segmentation_forward(enc, seg, *args)
    names = [n for n, _, _ in rec.records]
    assert any(n.startswith("pointmlpbf16") for n in names), names
    np.testing.assert_array_equal(enc.min_idx.cpu().numpy(), g["min_idx"])
    assert tuple(score.shape) == (B, 50, N)
    assert_close_rms(score.float().cpu().numpy(), g["score_segmenter"], BF16_TOL, "score_segmenter (node-wise layer 1, bf16)")
    assert_close_rms(score_dense.float().cpu().numpy(), g["score_segmenter"], BF16_TOL, "score_segmenter (dense layer 1, bf16)")


@pytest.mark.gpu
def test_autoencoder_forward_bf16(case="autoencoder_b2_n1024"):
    """Encoder -> decoder -> Chamfer loss under precision('bf16'): runs, predicted clouds within BF16_TOL of the reference."""
    from models import networks as NW, losses as LS
    from sonet_hip import ops, synth
    g = golden(case)
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.7, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40,
                    output_fc_pc_num=256, output_conv_pc_num=1024)
    enc, dec, crit = NW.Encoder(opt), NW.Decoder(opt), LS.ChamferLoss(opt)
    synth.fill_state_dict_(enc.state_dict(), seed)
    synth.fill_state_dict_(dec.state_dict(), seed + 1)
    enc.to(DEV).eval()
    dec.to(DEV).eval()
    with ops.precision("bf16"), torch.no_grad():
        feature = enc(cu(g["pc"]), cu(g["sn"]), cu(g["node"]), cu(g["node_knn_I"]), False, None)
        pred = dec(feature)
        loss = crit(pred.float(), cu(g["pc"])) + crit(dec.conv_pc4.float(), cu(g["pc"]))
    assert_close_rms(feature.float().cpu().numpy(), g["feature"], BF16_TOL, "feature (bf16)")
    assert_close_rms(pred.float().cpu().numpy(), g["predicted_pc"], BF16_TOL, "predicted_pc (bf16)")
    assert abs(float(loss) - float(g["loss"])) <= BF16_TOL * float(g["loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cout,Cin,L", [(3, 256, 128, 1504), (2, 64, 6, 776), (4, 128, 64, 64), (1, 512, 387, 1000), (2, 1024, 768, 136),
                                          (2, 384, 320, 3000), (3, 40, 17, 777), (1, 32, 32, 8),
                                          # >= 2048 units of 64 columns and L % 8 == 0: the streaming generation (LDS-DMA ring); partial row / column blocks, a short last unit
                                          (16, 256, 128, 15000), (64, 128, 64, 2048), (40, 64, 6, 5000), (36, 512, 387, 4000), (33, 100, 200, 4104)])
def test_wgrad_bf16_vs_float64(B, Cout, Cin, L):
    """sonet_wgrad_bf16 (one bf16 MFMA per product, f32 accumulation) == sum_b g[b] x[b]^T in float64 on the SAME bf16 operands, to f32
    rounding of the partial sums; ragged shapes (partial tiles, L not a multiple of 8 -> element-wise loads, short last unit)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B * 1000 + Cout + Cin + L)
    gg = (torch.randn(B, Cout, L, generator=g) * 1e-3).to(torch.bfloat16).to(DEV)
    xx = torch.randn(B, Cin, L, generator=g).to(torch.bfloat16).to(DEV)
    got = ops.wgrad_bf16(gg, xx)
    ref = torch.bmm(gg.double(), xx.double().transpose(1, 2)).sum(0)
    assert tuple(got.shape) == (Cout, Cin)
    assert_close_rms(got.cpu().numpy(), ref.cpu().numpy(), 2e-5, "wgrad bf16")
    assert torch.equal(got, ops.wgrad_bf16(gg, xx))               # deterministic (fixed-order reduction)


@pytest.mark.parametrize("B,C1,C2,Cout,L,M", [(64, 64, 256, 384, 15000, 64), (3, 64, 256, 384, 1000, 64), (2, 128, 0, 96, 130, 7),
                                               (5, 64, 0, 64, 2, 3), (2, 64, 64, 128, 4098, 255)])
def test_bf16_layer_with_pool_epilogue_equals_layer_then_index_max_gather(B, C1, C2, Cout, L, M):
    """sonet_pointmlp_bf16_pool (the layer's output pooled per node in its epilogue, never written) == sonet_pointmlp_bf16 followed by
    sonet_index_max_gather_bf16 on the stored tensor: positions and values bit for bit, incl. masked nodes, empty nodes, ids nobody owns,
    channels that never beat -1000 (position 0 and the value stored there) and ties (the first column wins)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C1 + Cout + L + M)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    if L >= 8:
        x1[:, :, 5] = x1[:, :, 3]                                 # ties between columns 3 and 5
        if x2 is not None:
            x2[:, :, 5] = x2[:, :, 3]
    W = (torch.randn(Cout, C1 + C2, generator=g) * (C1 + C2) ** -0.5).to(DEV)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    bias[1 % Cout] = -3000.0                                      # a channel that never beats -1000: position 0, the value stored there
    ids = torch.randint(0, M, (B, L), generator=g, dtype=torch.int32)
    if M > 3:
        ids[ids == 2] = 3                                          # node 2 is empty
    if L >= 8:
        ids[0, 5] = ids[0, 3]                                      # the tie inside one node
        ids[B - 1, :2] = torch.tensor([M + 5, -1], dtype=torch.int32)[:2]   # ids nobody owns
    ids = ids.to(DEV)
    row_max = torch.ones(B, M, dtype=torch.int32)
    row_max[:, 1 % M] = 0                                          # a masked node gathers position 0
    row_max = row_max.to(DEV)
    wp = ops.pointmlp_pack(W, "bf16")
    one = ops.const_vec(Cout, 1.0, DEV)
    assert ops.pointmlp_bf16_pool_ok(x1, x2, Cout, M)
    for rm in (row_max, None):
        idx, val = ops.pointmlp_bf16_pool(x1, wp, one, bias, False, Cout, ids, M, rm, x2=x2)
        y = ops.pointmlp(x1, wp, one, bias, False, Cout, x2=x2)
        idx_ref, val_ref = ops.index_max_gather(y, ids, M, rm)
        assert torch.equal(idx, idx_ref)
        assert torch.equal(val, val_ref)
    assert int((idx[:, 1 % Cout] != 0).sum()) == 0               # the -3000 channel: nothing beat -1000


def test_training_step_with_pool_epilogue_equals_the_storing_path():
    """The bf16 training step with the pool in the last layer's epilogue (first_pn_out never written) == the step that stores it and runs
    index_max on it: loss, every gradient, the updated BatchNorm statistics -- bit for bit (same positions, same values, same backward)."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 4, 1500
    res = {}
    with ops.precision("bf16"):
        for flag in (True, False):
            old = ops.POOLED_TRAIN_EPILOGUE
            ops.POOLED_TRAIN_EPILOGUE = flag
            try:
                opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True,
                                                                   feature_num=1024, activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3,
                                                                   som_k=9, som_k_type="avg", bn_momentum=0.1, bn_momentum_decay_step=None,
                                                                   bn_momentum_decay=0.6, classes=40)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False                      # (said explicitly: a live Segmenter of another test would keep the tensor)
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
                with ops.kernel_timing() as rec:
                    feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                    loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                    loss.backward()
                names = [n for n, _, _ in rec.records]
                assert any(n.startswith("pointmlpbf16_pool") for n in names) == flag
                assert any(n.startswith("index_max") for n in names) == (not flag)
                if flag:
                    with pytest.raises(RuntimeError, match="want_first_pn_out"):
                        enc.first_pn_out
                res[flag] = (loss.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                             {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone())
            finally:
                ops.POOLED_TRAIN_EPILOGUE = old
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][3], res[False][3])
    assert res[True][1].keys() == res[False][1].keys()
    for k in res[True][1]:
        assert torch.equal(res[True][1][k], res[False][1][k]), k
    for k in res[True][2]:
        assert torch.equal(res[True][2][k], res[False][2][k]), k
