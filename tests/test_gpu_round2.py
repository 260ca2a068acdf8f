"""GPU tests added in round 2: operand-range guard of the fp16-split arithmetic, the no-autograd fast path as the
reference's unmodified ``Model.test_model`` reaches it, BatchNorm running-statistics invalidation, the level-1 drop-in
(the reference's dense ``Encoder.forward`` data flow on the overlay operators), RCCL gradient all-reduce.

Float bound everywhere: |got-ref| <= 1e-5 * max(|ref|, rms(ref)); integer / index outputs bit-exact."""
import os
import subprocess
import sys
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close_rms, golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def make_opt(B, N, k=3, som_k=9, sn=True, classes=40):
    return Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=sn,
                     feature_num=1024, activation="relu", normalization="batch", dropout=0.7, node_num=64,
                     k=k, som_k=som_k, som_k_type="avg", bn_momentum=0.1, bn_momentum_decay_step=None,
                     bn_momentum_decay=0.6, classes=classes)


def build(B, N, seed=11, **kw):
    from models import networks as NW
    from sonet_hip import synth
    opt = make_opt(B, N, **kw)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    enc_sd = synth.fill_state_dict_(enc.state_dict(), seed)
    cls_sd = synth.fill_state_dict_(cls.state_dict(), seed + 1)
    return enc, cls, {k: v.clone() for k, v in enc_sd.items()}, {k: v.clone() for k, v in cls_sd.items()}


# ------------------------------------------------------------------------------------------ range guard: kernels
def _h3_layer(x, Cout=64, seed=0):
    from sonet_hip import ops
    g = torch.Generator().manual_seed(seed)
    W = (torch.randn(Cout, x.shape[1], generator=g) * 0.2).to(DEV)
    wp = ops.pointmlp_pack(W, "h3")
    one, zero = ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV)
    return ops.pointmlp(x, wp, one, zero, False, Cout), W


@pytest.mark.parametrize("what,expect", [("normal", None), ("big", "exceeds 2047"), ("neg_big", "exceeds 2047"), ("nan", "exceeds 2047"),
                                         ("neg_nan", "exceeds 2047"), ("inf", "exceeds 2047"), ("tiny", "below 2^-6"), ("zero", None)])
def test_h3_layer_reports_operand_range(what, expect):
    """Every h3 launch logs max |x| and max |w|; the scope names the launches that left the fp16-split range."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 48, 1000, generator=g)
    if what == "big":
        x[1, 7, 123] = 5000.0
    elif what == "neg_big":
        x[2, 47, 999] = -2048.5
    elif what == "nan":
        x[0, 0, 0] = float("nan")
    elif what == "neg_nan":
        x[0, 31, 17] = torch.tensor([0xFFC00000], dtype=torch.int64).to(torch.int32).view(torch.float32)[0]
    elif what == "inf":
        x[2, 5, 5] = float("-inf")
    elif what == "tiny":
        x = x * 1e-4
    elif what == "zero":
        x = torch.zeros_like(x)
    with ops.range_scope(DEV) as rs:
        _h3_layer(x.to(DEV))
    bad = rs.violations()
    if expect is None:
        assert bad == []
    else:
        assert len(bad) == 1 and expect in bad[0][1], bad
    # outside a scope nothing is logged and nothing breaks
    _h3_layer(x.to(DEV))
    torch.cuda.synchronize()


def test_h3_weight_range_is_logged():
    from sonet_hip import ops
    x = torch.randn(2, 32, 257).to(DEV)
    for scale, expect in ((1.0, None), (1e6, "max |w|"), (1e-4, "max |w|")):
        W = (torch.randn(64, 32) * 0.2 * scale).to(DEV)
        wp = ops.pointmlp_pack(W, "h3")
        with ops.range_scope(DEV) as rs:
            ops.pointmlp(x, wp, ops.const_vec(64, 1.0, DEV), ops.const_vec(64, 0.0, DEV), False, 64)
        bad = rs.violations()
        assert (bad == []) if expect is None else (len(bad) == 1 and expect in bad[0][1]), (scale, bad)


def test_fused_kernel_reports_hidden_activation_range():
    """The fused first PointNet logs |input|, |w| and the largest post-BatchNorm activation entering layers 2-4."""
    from models import layers as L
    from sonet_hip import ops, synth
    pr = L.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    x = torch.randn(2, 6, 700).to(DEV)
    with torch.no_grad():
        with ops.range_scope(DEV) as rs:
            pr(x)
        assert rs.violations() == []
        pr.layers[1].norm.weight[5] = 4.0e4                 # one channel of layer 2's BatchNorm blows up (in place: the version moves)
        with ops.range_scope(DEV) as rs:
            pr(x)
        bad = rs.violations()
        assert bad and "hidden activation" in bad[0][1] and bad[0][0].startswith("pointresnet_fused"), bad
        with ops.range_scope(DEV) as rs:
            pr(x * 1e4)
        assert any("max |x|" in b[1] for b in rs.violations())


# ------------------------------------------------------------------------------------------ range guard: model level
@pytest.mark.parametrize("case", ["act_1e4", "act_1e5_bn", "input_3e4", "nan_input", "weight_5e3"])
def test_encoder_out_of_range_activations_fall_back_to_x3(case):
    """Weights / inputs scaled so that activations leave the fp16-split range: the default (h3) forward must still match
    the oracle at 1e-5 -- through the guard's x3 recomputation -- and say so once."""
    from oracle import cpu_oracle as O
    from sonet_hip import ops, synth
    B, N = 2, 600
    enc, cls, enc_sd, cls_sd = build(B, N, seed=21)
    inp = synth.make_inputs(B, N, seed=9)
    if case == "act_1e4":                                    # hidden activations of the first PointNet reach ~1e4
        for sd in (enc.state_dict(), enc_sd):
            sd["first_pointnet.layers.0.norm.weight"].mul_(3.0e3)
            sd["first_pointnet.layers.1.conv.weight"].mul_(1.0 / 3.0e3)      # keeps the rest of the net at its usual scale
    elif case == "act_1e5_bn":                               # first_pn_out ~1e5 -> the KNN module's h3 layers see it
        for sd in (enc.state_dict(), enc_sd):
            sd["first_pointnet.layers.3.conv.weight"].mul_(2.0e4)
            sd["knnlayer.layers.0.conv.weight"][:, 3:].mul_(1.0 / 2.0e4)
    elif case == "weight_5e3":                               # weights of the fused kernel's last layer ~5e3: 32 w leaves fp16 (its limit is 2047)
        for sd in (enc.state_dict(), enc_sd):
            sd["first_pointnet.layers.3.conv.weight"].mul_(1.0e5)
            sd["knnlayer.layers.0.conv.weight"][:, 3:].mul_(1.0e-5)
    elif case == "input_3e4":                                # de-centred coordinates ~1e4
        for key in ("pc", "node"):
            inp[key] = inp[key] * 3.0e4
    elif case == "nan_input":
        inp["sn"][1, 2, 77] = float("nan")
    enc.to(DEV).eval()
    cls.to(DEV).eval()
    ops._range_warned = False
    old = ops.POINTMLP_PRECISION
    ops.POINTMLP_PRECISION = "h3"
    try:
        with torch.no_grad(), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            feat = enc(inp["pc"].to(DEV), inp["sn"].to(DEV), inp["node"].to(DEV), inp["node_knn_I"].to(DEV))
        assert any("x3" in str(x.message) for x in w), "the guard did not report the fallback"
    finally:
        ops.POINTMLP_PRECISION = old
    ref = O.encoder_forward(enc_sd, inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"])
    np.testing.assert_array_equal(enc.min_idx.cpu().numpy(), ref["min_idx"])
    got, want = feat.cpu().numpy(), ref["feature"].numpy()
    if case == "nan_input":                                  # (the per-node arg-max skips a NaN column, so the features are usually finite)
        assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN must be treated as in the reference"
        got, want = np.nan_to_num(got), np.nan_to_num(want)
    assert_close_rms(got, want, 1e-5, "feature (%s)" % case)
    assert_close_rms(enc.first_pn_out_masked_max.cpu().numpy(), ref["first_pn_out_masked_max"].numpy(), 1e-5, "masked_max (%s)" % case)


def test_encoder_in_range_forward_does_not_fall_back():
    from sonet_hip import ops, synth
    enc, cls, _, _ = build(2, 500)
    enc.to(DEV).eval()
    inp = synth.make_inputs(2, 500, seed=1, device=DEV)
    with torch.no_grad(), ops.kernel_timing() as rec:
        enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"])
    names = [n for n, _, _ in rec.records]
    assert sum(n.startswith("pointresnet_fused_pool") for n in names) == 1 and not any(n.startswith("pointmlpx3") for n in names), names


def test_graphed_forward_exposes_the_range_log():
    from sonet_hip import synth
    from sonet_hip.graph import GraphedForward
    enc, cls, _, _ = build(2, 400)
    enc.to(DEV).eval()
    cls.to(DEV).eval()
    inp = synth.make_inputs(2, 400, seed=2, device=DEV)
    args = (inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"])
    fwd = GraphedForward(lambda pc, sn, node, knn: cls(enc(pc, sn, node, knn)), tuple(t.clone() for t in args))
    fwd(*args)
    assert fwd.range_violations() == []
    big = (args[0] * 1.0e5, args[1], args[2] * 1.0e5, args[3])
    fwd(*big)
    assert fwd.range_violations(), "a replay on out-of-range data must show up in the log"
    fwd(*args)
    assert fwd.range_violations() == []


def test_graphed_forward_polls_the_range_log_without_being_asked():
    """A serving loop that only replays: the default periodic, non-blocking check raises within check_every (+ the copy's latency)
    replays of out-of-range data; in-range replays never raise."""
    from sonet_hip import synth
    from sonet_hip.graph import GraphedForward, CHECK_EVERY
    from sonet_hip.ops import SonetHipError
    enc, cls, _, _ = build(2, 400)
    enc.to(DEV).eval()
    cls.to(DEV).eval()
    inp = synth.make_inputs(2, 400, seed=2, device=DEV)
    args = (inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"])
    fwd = GraphedForward(lambda pc, sn, node, knn: cls(enc(pc, sn, node, knn)), tuple(t.clone() for t in args))
    assert fwd.check_every == CHECK_EVERY and CHECK_EVERY > 0
    for _ in range(3 * CHECK_EVERY):
        fwd(*args)
    torch.cuda.synchronize()
    fwd.poll_range()
    big = (args[0] * 1.0e5, args[1], args[2] * 1.0e5, args[3])
    with pytest.raises(SonetHipError, match="operand range"):
        for _ in range(4 * CHECK_EVERY):
            fwd(*big)
        torch.cuda.synchronize()
        fwd.poll_range()


def test_training_forward_range_violation_switches_to_x3_one_step_late():
    from sonet_hip import ops, synth
    enc, cls, _, _ = build(4, 300, seed=5)
    enc.to(DEV).train()
    inp = synth.make_inputs(4, 300, seed=3, device=DEV)
    old = ops.POINTMLP_PRECISION
    ops.POINTMLP_PRECISION, ops._range_warned = "h3", False
    ops._range_pending.clear()
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            enc(inp["pc"] * 1.0e5, inp["sn"], inp["node"] * 1.0e5, inp["node_knn_I"], is_train=True).sum().backward()
            torch.cuda.synchronize()
            assert ops.POINTMLP_PRECISION == "h3"                        # not looked at yet
            enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True)
            assert ops.POINTMLP_PRECISION == "x3" and any("x3" in str(x.message) for x in w)
    finally:
        ops.POINTMLP_PRECISION = old
        ops._range_pending.clear()


# ------------------------------------------------------------------------------------------ stale eval statistics
def test_running_stat_update_invalidates_the_folded_eval_affine():
    """A train-mode forward that changes ONLY the running statistics (no optimizer step) must be seen by the next eval."""
    from models import layers as L
    from sonet_hip import synth
    layer = L.EquivariantLayer(16, 32, "relu", "batch", 0.5)
    synth.fill_state_dict_(layer.state_dict(), 3)
    layer.to(DEV)
    x = (torch.randn(4, 16, 300) * 3 + 1).to(DEV)
    with torch.no_grad():
        layer.eval()
        y0 = layer(x).clone()
        layer.train()
        layer(x)                                             # updates running_mean / running_var through the kernel
        layer.eval()
        y1 = layer(x)
        bn = layer.norm
        ref = torch.relu(torch.nn.functional.batch_norm(torch.nn.functional.conv1d(x, layer.conv.weight, layer.conv.bias),
                                                        bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
    assert not torch.allclose(y0, y1), "the statistics did change"
    assert_close_rms(y1.cpu().numpy(), ref.cpu().numpy(), 1e-5, "eval after a stats-only update")
    # ... and the fused first PointNet keys on the same thing
    pr = L.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.5, None, 1)
    synth.fill_state_dict_(pr.state_dict(), 4)
    pr.to(DEV)
    xp = torch.randn(2, 6, 500).to(DEV)
    with torch.no_grad():
        pr.eval()
        a = pr(xp).clone()
        pr.train()
        pr(xp)
        pr.eval()
        b = pr(xp)
        pr.layers[3].conv.bias.add_(1.0)                    # bias-only change of the norm-free last layer
        c = pr(xp)
    assert not torch.allclose(a, b)
    assert_close_rms((c - b).cpu().numpy(), np.ones(tuple(b.shape), dtype=np.float32), 1e-4, "bias-only change reaches the fused kernel")


# ------------------------------------------------------------------------------------------ the reference's callers, unmodified
def test_reference_shaped_test_model_reaches_the_fast_path():
    """models/classifier.py:101-105 runs eval WITHOUT torch.no_grad(): ``encoder.eval(); forward(is_train=False)``.
    That call must take the fused no-autograd kernels and give the same numbers as under no_grad."""
    from sonet_hip import ops, synth
    enc, cls, _, _ = build(2, 700, seed=13)
    enc.to(DEV)
    cls.to(DEV)
    inp = synth.make_inputs(2, 700, seed=4, device=DEV)
    args = (inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"])

    class Model:                                             # the shape of the reference's task Model (classifier.py:74-76,101-105)
        def forward(self, is_train=False, epoch=None):
            self.feature = enc(*args, is_train, epoch)
            self.score = cls(self.feature, epoch)

        def test_model(self):
            enc.eval()
            cls.eval()
            self.forward(is_train=False)
            self.loss = torch.nn.functional.cross_entropy(self.score, inp["label"])

    m = Model()
    assert torch.is_grad_enabled()
    with ops.kernel_timing() as rec:
        m.test_model()
    names = [n for n, _, _ in rec.records]
    assert any(n.startswith("pointresnet_fused_pool") for n in names), names
    assert sum(n.startswith("linear_act") for n in names) == 3, names
    assert not any(n.startswith("index_max") for n in names)
    assert not m.score.requires_grad and torch.isfinite(m.loss)
    with torch.no_grad():
        ref = cls(enc(*args, is_train=False))
    assert torch.equal(ref, m.score)
    # opting out restores autograd through an eval-mode forward (fine-tuning with frozen BatchNorm statistics)
    enc.inference = False
    feat = enc(*args, is_train=False)
    assert feat.requires_grad
    feat.sum().backward()
    assert enc.first_pointnet.layers[0].conv.weight.grad is not None
    assert_close_rms(feat.detach().cpu().numpy(), m.feature.cpu().numpy(), 1e-5, "autograd eval vs fast path")
    # a training call is untouched
    enc.inference = True
    enc.train()
    assert enc(*args, is_train=True).requires_grad


def test_reference_shaped_segmenter_test_model_reaches_the_fast_path():
    """models/segmenter.py:79-109,126-128 (eval without no_grad): mask argmax -> three gathers -> Segmenter, unmodified."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 2, 384
    opt = make_opt(B, N, classes=50)
    enc, seg = NW.Encoder(opt), NW.Segmenter(opt)
    synth.fill_state_dict_(enc.state_dict(), 31)
    synth.fill_state_dict_(seg.state_dict(), 32)
    enc.to(DEV).eval()
    seg.to(DEV).eval()
    inp = synth.make_inputs(B, N, seed=6, device=DEV)
    label = torch.randint(0, 16, (B,), device=DEV)

    def reference_model_forward():
        feature = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], False, None)
        idx = torch.max(enc.mask, dim=2)[1].unsqueeze(1)
        kN = idx.shape[2]
        g1 = torch.gather(enc.first_pn_out_masked_max, 2, idx.expand(B, 384, kN))
        g2 = torch.gather(enc.knn_feature_1, 2, idx.expand(B, 512, kN))
        g3 = torch.gather(enc.final_pn_out, 2, idx.expand(B, 1024, kN))
        return seg(enc.x_decentered, inp["pc"], enc.centers, inp["sn"], label, enc.first_pn_out, g1, g2, g3, feature)

    enc.want_first_pn_out = True
    with ops.kernel_timing() as rec:
        score = reference_model_forward()
    names = [n for n, _, _ in rec.records]
    assert not score.requires_grad
    assert any(n.startswith("pointresnet_fused_L") for n in names), names       # one-kernel first PointNet (store variant)
    with torch.no_grad():
        ref = reference_model_forward()
    assert torch.equal(ref, score)
    with torch.no_grad():
        nodewise = NW.segmentation_forward(enc, seg, inp["pc"], inp["sn"], label, inp["node"], inp["node_knn_I"])
    assert_close_rms(nodewise.cpu().numpy(), score.cpu().numpy(), 1e-5, "node-wise layer 1 vs the dense head")


# ------------------------------------------------------------------------------------------ level 1 on hardware
def level1_encoder_forward(enc, x, sn, node, node_knn_I):
    """The reference's DENSE Encoder.forward data flow (models/networks.py:124-199), restated on the level-1 drop-in
    operators only: BatchSOM.query_topk's one-hot mask, masked mean through a B x 3 x kN x M product, centres as a
    mask-weighted node sum, index_max.forward_cuda + Tensor.gather, the KNN module and the final PointNet as modules."""
    import index_max
    opt = enc.opt
    sb = enc.som_builder
    sb.node = node.clone()                                                              # :124
    mask, row_max, min_idx = sb.query_topk(x, k=opt.k)                                  # :127   B x kN x M, B x M, B x kN
    counts = mask.sum(dim=1)                                                            # :128
    maskf = mask.unsqueeze(1).float()                                                   # B x 1 x kN x M
    xs, sns = x.repeat(1, 1, opt.k), sn.repeat(1, 1, opt.k)                             # :131-137
    mean = (xs.unsqueeze(3) * maskf).sum(dim=2) / (counts.unsqueeze(1).float() + 1e-5)  # :140-142
    sb.node = mean                                                                      # :143
    centers = (maskf * mean.unsqueeze(2)).sum(dim=3)                                    # :168-169
    xdec = xs - centers                                                                 # :171
    first = enc.first_pointnet(torch.cat((xdec, sns), dim=1))                           # :172-176
    M = node.shape[2]
    gi = index_max.forward_cuda(first.detach(), min_idx.int(), M).long()                # :180-184
    pooled = first.gather(dim=2, index=gi * row_max.unsqueeze(1).long())                # :185
    center1, feat1 = enc.knnlayer(mean, pooled, node_knn_I, opt.som_k, opt.som_k_type)  # :189
    final = enc.final_pointnet(torch.cat((center1, feat1), dim=1))                      # :192
    return dict(min_idx=min_idx, mask=mask, som_node=mean, centers=centers, x_decentered=xdec, first_pn_out=first,
                first_pn_out_masked_max=pooled, knn_center_1=center1, knn_feature_1=feat1, final_pn_out=final,
                feature=final.max(dim=2)[0])                                            # :197


@pytest.mark.parametrize("case", ["classifier_b8_n1024", "classifier_b2_n5000"])
@pytest.mark.parametrize("mode", ["h3", "x3", "f32"])
def test_level1_dense_encoder_flow_on_the_overlay_ops_golden(case, mode):
    """both drop-in levels must pass the same oracle tests.  Level 1 = the reference's own forward on top of
    the new index_max / BatchSOM / layers / operations; checked here against the live-reference fixture."""
    from models import networks as NW
    from sonet_hip import ops, synth
    g = golden(case)
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    opt = make_opt(B, N, k=int(g["k"]), som_k=int(g["som_k"]))
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), seed)
    synth.fill_state_dict_(cls.state_dict(), seed + 1)
    enc.to(DEV).eval()
    cls.to(DEV).eval()
    with ops.precision(mode), torch.no_grad(), ops.kernel_timing() as rec:
        r = level1_encoder_forward(enc, cu(g["pc"]), cu(g["sn"]), cu(g["node"]), cu(g["node_knn_I"]))
        score = cls(r["feature"])
    names = [n for n, _, _ in rec.records]
    assert "index_max" in names and "som_mask" in names and "som_assign" in names, names
    np.testing.assert_array_equal(r["min_idx"].cpu().numpy(), g["min_idx"])
    np.testing.assert_array_equal(r["mask"].sum(1).cpu().numpy(), g["mask_row_sum"])
    tol = 1e-5
    assert_close_rms(r["som_node"].cpu().numpy(), g["som_node"], tol, "som_node")
    assert_close_rms(r["centers"][:, :, ::7].cpu().numpy(), g["centers"], tol, "centers")
    assert_close_rms(r["x_decentered"][:, :, ::7].cpu().numpy(), g["x_decentered"], tol, "x_decentered")
    assert_close_rms(r["first_pn_out"][:, ::16, ::5].cpu().numpy(), g["first_pn_out_sub"], tol, "first_pn_out")
    assert_close_rms(r["first_pn_out_masked_max"].cpu().numpy(), g["first_pn_out_masked_max"], tol, "masked_max")
    assert_close_rms(r["knn_feature_1"][:, ::4].cpu().numpy(), g["knn_feature_1"], tol, "knn_feature_1")
    assert_close_rms(r["final_pn_out"][:, ::4].cpu().numpy(), g["final_pn_out"], tol, "final_pn_out")
    assert_close_rms(r["feature"].cpu().numpy(), g["feature"], tol, "feature")
    assert_close_rms(score.cpu().numpy(), g["score"], tol, "score")


# ------------------------------------------------------------------------------------------ RCCL
def test_gradient_all_reducer_on_rccl():
    """GradientAllReducer with backend nccl (= RCCL) at world_size = the visible GPUs (1 is allowed: the process group,
    the hooks and the collective calls are the same code; 2+ also checks the averaged values)."""
    n = torch.cuda.device_count()
    script = os.path.join(ROOT, "tests", "dp_rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", "29631", script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RCCL_OK world=%d" % n in r.stdout, r.stdout[-2000:]


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0 and "only %d GPU" % n in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,M,L,NL", [(3, 512, 64, 576, 3), (2, 37, 10, 33, 3), (1, 8, 5, 7, 0), (2, 64, 100, 1000, 4)])
def test_node_gather_lead_affine_act_vs_torch(B, C, M, L, NL):
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B * C + L)
    z = torch.randn(B, C, M, generator=g)
    gidx = torch.randint(-1, M + 1, (B, L), generator=g, dtype=torch.int32)
    lead = torch.randn(B, NL, L, generator=g)
    wl = torch.randn(C, NL, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ok = (gidx >= 0) & (gidx < M)
    zg = torch.gather(z.double(), 2, gidx.clamp(0, M - 1).long().unsqueeze(1).expand(B, C, L)) * ok.unsqueeze(1)
    ref = torch.relu((zg + torch.einsum("ci,bil->bcl", wl.double(), lead.double())) * sc.double().view(1, -1, 1) + sh.double().view(1, -1, 1))
    got = ops.node_gather_lead_affine_act(cu(z), cu(gidx), cu(lead), cu(wl), cu(sc), cu(sh), True).cpu().double()
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_knn_module_linear_split_matches_gathered_layer():
    """KNNModule layer 1 as (layer on the node features) + gather + coordinate channels == the layer over the gathered columns,
    within the split arithmetic's own noise; the encoder forward meets the oracle on both (the golden tests run the default)."""
    from models import layers as Lm
    from sonet_hip import ops, synth
    B, M, K, C = 4, 64, 9, 384
    knn = Lm.KNNModule(3 + C, (512, 512), "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(knn.state_dict(), seed=3)
    knn.to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    coord, x = cu(torch.rand(B, 3, M, generator=g) * 2 - 1), cu(torch.randn(B, C, M, generator=g))
    knn_I = ops.knn_self(coord, K)
    out = {}
    old = ops.NODE_LINEAR_SPLIT
    try:
        for flag in (True, False):
            ops.NODE_LINEAR_SPLIT = flag
            with torch.no_grad(), ops.kernel_timing() as rec:
                out[flag] = knn(coord, x, knn_I, K, "center")[1]
            names = [n for n, _, _ in rec.records]
            assert any(n.startswith("node_gather_lead") for n in names) == flag, names
    finally:
        ops.NODE_LINEAR_SPLIT = old
    err = (out[True] - out[False]).abs().max()
    assert float(err) <= 1e-5 * max(1.0, float(out[False].abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cout,Cin,L", [(3, 256, 128, 1500), (2, 64, 6, 777), (4, 128, 64, 64), (1, 512, 387, 1000), (2, 1024, 768, 130),
                                         (5, 96, 33, 1), (2, 256, 320, 4099)])
def test_wgrad_x3_vs_float64(B, Cout, Cin, L):
    """sonet_wgrad_x3_f32 == sum_b g[b] x[b]^T in float64 at f32-class accuracy (operands of very different scale: gradients
    ~1e-4, activations ~1), ragged L, channel counts that are no multiple of the 32-row tile, several column slices; and
    bitwise reproducible (fixed summation order)."""
    from sonet_hip import ops
    gen = torch.Generator().manual_seed(Cout + Cin + L)
    g = cu(torch.randn(B, Cout, L, generator=gen) * 1e-4 * (1 + 10 * torch.rand(1, Cout, 1, generator=gen)))
    x = cu(torch.randn(B, Cin, L, generator=gen) * (0.1 + torch.rand(1, Cin, 1, generator=gen)))
    ref = torch.einsum("bol,bcl->oc", g.double(), x.double())
    got = ops.wgrad_x3(g, x)
    assert tuple(got.shape) == (Cout, Cin)
    scale = float(torch.einsum("bol,bcl->oc", g.double().abs(), x.double().abs()).max())       # the sum's own conditioning
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale
    assert torch.equal(got, ops.wgrad_x3(g, x))
    f32 = torch.bmm(g, x.transpose(1, 2)).sum(0)
    assert float((got.double() - ref).abs().max()) <= 4.0 * float((f32.double() - ref).abs().max()) + 1e-7 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["h3", "x3"])
@pytest.mark.parametrize("B,C1,C2,Cout,L", [(3, 64, 0, 128, 1500), (2, 6, 0, 64, 777), (2, 256, 64, 384, 300), (5, 128, 0, 256, 33)])
def test_pointmlp_statistics_epilogue(mode, B, C1, C2, Cout, L):
    """The layer kernel's statistics epilogue: same output as the plain launch bit for bit, mean / biased variance equal to the
    separate pass (f64 sums over every element) to 1e-6 of the channel's scale -- with a large common offset (mean^2 >> var)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = cu(torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5)
    x1 = cu(torch.randn(B, C1, L, generator=g))
    x2 = cu(torch.randn(B, C2, L, generator=g)) if C2 else None
    scale = cu(torch.ones(Cout))
    shift = cu(torch.randn(Cout, generator=g) * 3.0)                       # the conv bias: a common offset per channel
    wp = ops.pointmlp_pack(w, mode)
    y0 = ops.pointmlp(x1, wp, scale, shift, False, Cout, x2=x2)
    m0, v0 = ops.channel_stats(y0)
    y1, m1, v1 = ops.pointmlp_stats(x1, wp, scale, shift, False, Cout, x2=x2)
    assert torch.equal(y0, y1)
    ref = y0.double()
    mref, vref = ref.mean(dim=(0, 2)), ref.var(dim=(0, 2), unbiased=False)
    sc = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
    assert float(((m1.double().cpu() - mref.cpu()).abs() / sc.cpu()).max()) < 1e-6
    assert float(((v1.double().cpu() - vref.cpu()).abs() / (sc.cpu() ** 2)).max()) < 2e-6
    assert float(((m0.double() - m1.double()).abs() / sc).max()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,C1,C2,Cout,L,M", [(3, 384, 9, 1024, 700, 64), (2, 64, 0, 256, 33, 5), (2, 128, 6, 512, 3072, 64)])
def test_pointmlp_nodeadd_epilogue(B, C1, C2, Cout, L, M):
    """Layer with a per-node addend gathered in the epilogue == the plain layer (unit scale) + sonet_node_add_affine_act_f32, to the
    rounding of the intermediate tensor; and == float64."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = cu(torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5)
    x1 = cu(torch.randn(B, C1, L, generator=g))
    x2 = cu(torch.randn(B, C2, L, generator=g)) if C2 else None
    z = cu(torch.randn(B, Cout, M, generator=g))
    idx = cu(torch.randint(-1, M + 1, (B, L), generator=g, dtype=torch.int32))
    scale, shift = cu(torch.rand(Cout, generator=g) + 0.5), cu(torch.randn(Cout, generator=g))
    wp = ops.pointmlp_pack(w, "h3")
    got = ops.pointmlp_nodeadd(x1, wp, scale, shift, True, Cout, z, idx, x2=x2)
    t = ops.pointmlp(x1, wp, torch.ones_like(scale), torch.zeros_like(shift), False, Cout, x2=x2)
    two = ops.node_add_affine_act_(t, z, idx, scale, shift, True)
    xin = torch.cat([x1, x2], 1).double() if x2 is not None else x1.double()
    ok = (idx >= 0) & (idx < M)
    zg = torch.gather(z.double(), 2, idx.clamp(0, M - 1).long().unsqueeze(1).expand(B, Cout, L)) * ok.unsqueeze(1)
    ref = torch.relu((torch.einsum("oc,bcl->bol", w.double(), xin) + zg) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1))
    bound = 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((got.double() - ref).abs().max()) <= bound
    assert float((got - two).abs().max()) <= 4e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_two_graphed_forwards_in_flight_on_two_streams():
    """bench.py's default launch mode: independent batches captured into separate HIP graphs and replayed round-robin on separate
    streams.  Every replay must equal the same graph replayed alone, also after eager allocations in between (with hipMemsetAsync
    NODES in the graphs this faulted intermittently on ROCm 7.2 -- the forward now clears its workspaces with a kernel)."""
    from models import networks as NW
    from sonet_hip import synth
    from sonet_hip.graph import GraphedForward
    import bench
    dev = torch.device(DEV)
    B, N, P = 8, 2000, 2
    opt = bench.make_opt(dev, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), 0)
    synth.fill_state_dict_(cls.state_dict(), 1)
    enc.to(dev).eval()
    cls.to(dev).eval()
    inps = [synth.make_inputs(B, N, seed=7 + q, device=dev) for q in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    with torch.no_grad():
        graphs = [GraphedForward(lambda pc, sn, node, knn: cls(enc(pc, sn, node, knn, is_train=False)),
                                 (i["pc"], i["sn"], i["node"], i["node_knn_I"]), warmup=2) for i in inps]

        def rep(q):
            with torch.cuda.stream(streams[q]):
                i = inps[q]
                return graphs[q](i["pc"], i["sn"], i["node"], i["node_knn_I"])
        alone = []
        for q in range(P):
            rep(q)
            torch.cuda.synchronize()
            alone.append(graphs[q].static_output.clone())
        for rnd in range(6):
            for s in range(8):
                rep((rnd + s) % P)
            torch.cuda.synchronize()
            for q in range(P):
                assert torch.equal(graphs[q].static_output, alone[q]), (rnd, q)
            junk = [torch.isfinite(graphs[q].static_output).all().item() for q in range(P)]      # eager allocations between the rounds
            assert all(junk)
        eager = cls(enc(inps[1]["pc"], inps[1]["sn"], inps[1]["node"], inps[1]["node_knn_I"], is_train=False))
        assert torch.equal(eager, alone[1])


def test_h3_per_channel_range_case():
    """The adversarial case of the per-launch range guard: ONE input channel 1e-4 x the tensor maximum, met by weights 1e4 x larger in
    that column.  The launch's max |x| is fine, the small channel's fp16 residual is not.  The guard's weight side (ops.h3_weight_ok:
    input columns more than 128 x apart -> x3 pack) catches it: the layer still meets 1e-5 against float64; the raw h3 kernel on the
    same operands is printed for reference (it does not)."""
    import warnings
    from models import layers as Lm
    from sonet_hip import ops
    g = torch.Generator().manual_seed(11)
    B, Cin, Cout, L = 2, 64, 128, 2000
    x = torch.randn(B, Cin, L, generator=g)
    x[:, 5] *= 1e-4
    lyr = Lm.EquivariantLayer(Cin, Cout, activation=None, normalization=None)
    with torch.no_grad():
        lyr.conv.weight.normal_(0, 0.1, generator=g)
        lyr.conv.weight[:, 5] *= 1e4
        lyr.conv.bias.zero_()
    lyr.to(DEV).eval()
    ref = torch.einsum("oc,bcl->bol", lyr.conv.weight.detach().double().cpu().reshape(Cout, Cin), x.double()).numpy()
    assert not ops.h3_weight_ok(lyr.conv.weight.reshape(Cout, Cin))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with torch.no_grad(), ops.precision("h3"), ops.kernel_timing() as rec:
            y = lyr(x.to(DEV))
    names = [n for n, _, _ in rec.records]
    assert any(n.startswith("pointmlpx3") for n in names) and not any(n.startswith("pointmlph3") for n in names), names
    assert_close_rms(y.cpu().numpy(), ref, 1e-5, "layer with a 1e-4 channel and 1e4 weights (guarded)")
    wp = ops.pointmlp_pack(lyr.conv.weight.detach().reshape(Cout, Cin).contiguous(), "h3")
    raw = ops.pointmlp(x.to(DEV), wp, ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV), False, Cout).cpu().double().numpy()
    rms = np.sqrt(np.mean(ref ** 2))
    print("raw h3 on the adversarial operands: worst err / (1e-5 max(|ref|, rms)) = %.2f" % float((np.abs(raw - ref) / (1e-5 * np.maximum(np.abs(ref), rms))).max()))
    # ordinary weights pass the column test (and keep the fp16 split)
    assert ops.h3_weight_ok(torch.randn(256, 128, generator=g).to(DEV) * 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["x3", "h3", "bf16"])
@pytest.mark.parametrize("Cout,Cin,lo,Ci,Cp", [(512, 387, 0, 387, 512), (384, 320, 64, 256, 256), (384, 320, 0, 64, 64), (1024, 768, 0, 768, 768),
                                              (768, 515, 0, 515, 640), (128, 64, 0, 64, 64)])
def test_transposed_pack_read_in_place_equals_the_pack_of_the_transposed_copy(mode, Cout, Cin, lo, Ci, Cp):
    """sonet_pointmlp_{x3,h3,bf16}_pack_strided on W (the dgrad's weights W[:, lo:lo+Ci]^T, zero rows up to Cp) == the pack of the
    materialised transposed, padded copy: byte for byte (range trailer included)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(Cout + Cin + lo)
    W = (torch.randn(Cout, Cin, generator=g) * 0.2).to(DEV)
    wt = W[:, lo:lo + Ci].t().contiguous()
    if Cp != Ci:
        wt = torch.cat((wt, wt.new_zeros(Cp - Ci, Cout)), dim=0)
    ref = ops.pointmlp_pack(wt, mode)
    got = ops.pointmlp_pack_transposed(W, lo, Ci, Cp, mode)
    assert got.dtype == ref.dtype and got.numel() == ref.numel() and torch.equal(got, ref)
