"""The heads' B x C layers in training (``sonet_fc_bn_act_fwd_f32`` / ``_bwd_f32`` / ``sonet_fc_dx_f32`` behind models.layers.MyLinear)
against the aten path of the same module (nn.Linear + F.batch_norm + ReLU, autograd) and against float64."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,Cin,Cout,norm,act", [(64, 1024, 512, "batch", "relu"), (64, 512, 256, "batch", "relu"), (64, 256, 40, None, None),
                                                  (8, 1024, 512, "batch", "relu"), (2, 64, 8, "batch", "relu"), (100, 20, 12, "batch", None),
                                                  (128, 132, 36, None, "relu"), (1, 256, 40, None, None), (65, 68, 4, "batch", "relu")])
def test_fc_layer_training_kernels_vs_aten_and_float64(B, Cin, Cout, norm, act):
    from models import layers as L
    from sonet_hip import ops
    torch.manual_seed(B * 7 + Cin + Cout)
    own = L.MyLinear(Cin, Cout, activation=act, normalization=norm).to(DEV).train()
    with torch.no_grad():
        own.linear.bias.normal_(0, 0.1)
        if norm:
            own.norm.weight.uniform_(0.5, 1.5)
            own.norm.bias.normal_(0, 0.2)
            own.norm.running_mean.normal_(0, 0.3)
            own.norm.running_var.uniform_(0.5, 2.0)
    ref, r64 = copy.deepcopy(own), copy.deepcopy(own).double()
    x = torch.randn(B, Cin, device=DEV)
    gy = torch.randn(B, Cout, device=DEV)
    xa, xb, xc = x.clone().requires_grad_(True), x.clone().requires_grad_(True), x.double().requires_grad_(True)
    assert ops.fc_head_ok(xa, own.linear.weight)
    own.eval()
    with torch.no_grad():
        own(x, 0)                                                   # (fills the folded eval-mode affine: it must not survive the training step)
    own.train()
    with ops.kernel_timing() as rec:
        ya = own(xa, 0)
        (ya * gy).sum().backward()
    names = [n for n, _, _ in rec.records]
    assert any(n.startswith("fc_bn_act_fwd") for n in names) and any(n.startswith("fc_bn_act_bwd") for n in names) and any(n.startswith("fc_dx") for n in names), names
    old = ops.FC_HEAD
    ops.FC_HEAD = False
    try:
        yb = ref(xb, 0)
        (yb * gy).sum().backward()
        yc = r64(xc, 0)
        (yc * gy.double()).sum().backward()
    finally:
        ops.FC_HEAD = old
    # as close to float64 as aten is (within a factor), and close to aten in absolute terms
    def chk(a, b, c, what):
        ea, eb = _rel(a, c), _rel(b, c)
        assert ea <= max(4.0 * eb, 2e-6), (what, ea, eb)
    chk(ya, yb, yc, "y")
    chk(xa.grad, xb.grad, xc.grad, "dx")
    chk(own.linear.weight.grad, ref.linear.weight.grad, r64.linear.weight.grad, "dW")
    if norm:
        chk(own.norm.weight.grad, ref.norm.weight.grad, r64.norm.weight.grad, "dgamma")
        chk(own.norm.bias.grad, ref.norm.bias.grad, r64.norm.bias.grad, "dbeta")
        chk(own.norm.running_mean, ref.norm.running_mean, r64.norm.running_mean, "running_mean")
        chk(own.norm.running_var, ref.norm.running_var, r64.norm.running_var, "running_var")
        # a bias in front of BatchNorm: its gradient is rounding noise around zero in every implementation
        assert float(own.linear.bias.grad.abs().max()) <= 1e-4 * float(gy.abs().sum(0).max())
    else:
        chk(own.linear.bias.grad, ref.linear.bias.grad, r64.linear.bias.grad, "dbias")
    # eval after the training step: the folded affine must see the running statistics the kernel just wrote
    if norm:
        own.eval(), ref.eval()
        with torch.no_grad():
            ea, eb = own(x, 0), ref(x, 0)
        assert _rel(ea, eb) <= 1e-5, _rel(ea, eb)
        own.train(), ref.train()
    # deterministic
    own.zero_grad(set_to_none=True)
    xa2 = x.clone().requires_grad_(True)
    with torch.no_grad():
        if norm:
            own.norm.running_mean.copy_(ref.norm.running_mean)
    y2 = own(xa2, 0)
    (y2 * gy).sum().backward()
    assert torch.equal(y2, ya) and torch.equal(xa2.grad, xa.grad)


def test_fc_layer_shapes_outside_the_kernels_stay_on_aten():
    from models import layers as L
    from sonet_hip import ops
    lay = L.MyLinear(30, 12, activation="relu", normalization="batch").to(DEV).train()          # 30 input channels: not a multiple of 4
    x = torch.randn(16, 30, device=DEV, requires_grad=True)
    assert not ops.fc_head_ok(x, lay.linear.weight)
    with ops.kernel_timing() as rec:
        lay(x, 0).sum().backward()
    assert not any(n.startswith("fc_") for n, _, _ in rec.records)
    wide = L.MyLinear(32, 12, activation="relu", normalization="batch").to(DEV).train()
    xw = torch.randn(129, 32, device=DEV, requires_grad=True)                                     # more rows than the kernels take
    assert not ops.fc_head_ok(xw, wide.linear.weight)
    wide(xw, 0).sum().backward()
    assert xw.grad is not None
    one = L.MyLinear(32, 12, activation="relu", normalization="batch").to(DEV).train()
    with pytest.raises(ValueError):                                                               # one row, training BatchNorm: as F.batch_norm
        one(torch.randn(1, 32, device=DEV, requires_grad=True), 0)


def test_fc_entry_points_reject_what_they_cannot_run():
    from sonet_hip import ops
    x = torch.randn(8, 32, device=DEV)
    w = torch.randn(6, 32, device=DEV)                                                            # 6 output channels
    with pytest.raises(ops.SonetHipError):
        ops.fc_bn_act_fwd(x, w, None, None, None, None, None, 0.1, 1e-5, True)
    w = torch.randn(8, 32, device=DEV)
    with pytest.raises(ops.SonetHipError):                                                        # one row with batch statistics
        ops.fc_bn_act_fwd(x[:1].contiguous(), w, None, torch.ones(8, device=DEV), torch.zeros(8, device=DEV), None, None, 0.1, 1e-5, True)


def test_classifier_head_training_step_matches_aten():
    """The whole classifier head (1024 -> 512 -> 256 -> 40, BatchNorm + ReLU, cross entropy): loss, every gradient and the running
    statistics against the aten path of the same modules."""
    from argparse import Namespace

    from models import networks as NW
    from sonet_hip import ops
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=64, input_pc_num=5000, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    torch.manual_seed(3)
    a = NW.Classifier(opt).to(DEV).train()
    b = copy.deepcopy(a)
    feat = torch.randn(64, 1024, device=DEV)
    label = torch.randint(0, 40, (64,), device=DEV)
    fa, fb = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    la = torch.nn.functional.cross_entropy(a(fa, 0), label)
    la.backward()
    old = ops.FC_HEAD
    ops.FC_HEAD = False
    try:
        lb = torch.nn.functional.cross_entropy(b(fb, 0), label)
        lb.backward()
    finally:
        ops.FC_HEAD = old
    assert abs(float(la.detach()) - float(lb.detach())) <= 1e-5 * abs(float(lb.detach()))
    assert _rel(fa.grad, fb.grad) <= 1e-4
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if float(q.grad.abs().max()) < 1e-6:                                                     # (a bias in front of a BatchNorm)
            assert float(p.grad.abs().max()) < 1e-5, k
            continue
        assert _rel(p.grad, q.grad) <= 1e-4, (k, _rel(p.grad, q.grad))
    for (k, p), (_, q) in zip(a.named_buffers(), b.named_buffers()):
        if p.dtype.is_floating_point:
            assert _rel(p, q) <= 1e-5, k
