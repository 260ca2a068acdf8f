"""bf16 training forward with NORMALISE-ON-LOAD (round 6): the consumers of a hidden layer's output read its RAW (bf16) output and apply
act(raw * scale + shift) in their operand loads; the normalised tensor is never written (models/layers.py:60-70, :282-296 forward;
SURVEY.md 7, hard part 3).  Every kernel must reproduce, bit for bit, the same kernel fed with what sonet_channel_affine_act_bf16 stores."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _raw_case(B, C, L, seed):
    g = torch.Generator().manual_seed(seed)
    raw = (torch.randn(B, C, L, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    sc = (torch.rand(C, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(C, generator=g) * 0.4).to(DEV)
    return raw, sc, sh


@pytest.mark.parametrize("B,C1,C2,Cout,L", [(64, 64, 0, 128, 15000), (64, 128, 0, 256, 15000), (40, 64, 256, 384, 14002), (33, 64, 64, 64, 16384)])
def test_bf16_layer_with_normalise_on_load_equals_layer_on_the_normalised_tensor(B, C1, C2, Cout, L):
    """sonet_pointmlp_bf16_stats_xaff on RAW inputs == sonet_pointmlp_bf16_stats on sonet_channel_affine_act_bf16's output: the stored
    (bf16) outputs bit for bit, the batch statistics too (same kernel, same partial sums)."""
    from sonet_hip import ops
    r1, s1, h1 = _raw_case(B, C1, L, 1 + C1 + L)
    r2, s2, h2 = _raw_case(B, C2, L, 2 + C2 + L) if C2 else (None, None, None)
    gen = torch.Generator().manual_seed(Cout)
    W = (torch.randn(Cout, C1 + C2, generator=gen) * (C1 + C2) ** -0.5).to(DEV)
    bias = (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
    with ops.precision("bf16"):
        wp = ops.pointmlp_pack(W, "bf16")
        one = ops.const_vec(Cout, 1.0, DEV)
        y1 = ops.channel_affine_act(r1, s1, h1, True)
        y2 = ops.channel_affine_act(r2, s2, h2, False) if C2 else None
        ref = ops.pointmlp_stats(y1, wp, one, bias, False, Cout, x2=y2)
        xa = (s1, h1, True) + ((s2, h2, False) if C2 else ())
        got = ops.pointmlp_stats(r1, wp, one, bias, False, Cout, x2=r2, xaff=xa)
    assert got[0].dtype == torch.bfloat16
    assert torch.equal(got[0].view(torch.int16), ref[0].view(torch.int16))
    assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])


@pytest.mark.parametrize("B,C1,C2,Cout,L,M", [(64, 64, 256, 384, 15000, 64), (5, 64, 0, 64, 1002, 7), (3, 128, 64, 96, 4098, 255)])
def test_bf16_pool_with_normalise_on_load_equals_pool_on_the_normalised_tensors(B, C1, C2, Cout, L, M):
    from sonet_hip import ops
    r1, s1, h1 = _raw_case(B, C1, L, 5 + L)
    r2, s2, h2 = _raw_case(B, C2, L, 6 + L) if C2 else (None, None, None)
    gen = torch.Generator().manual_seed(Cout + M)
    W = (torch.randn(Cout, C1 + C2, generator=gen) * (C1 + C2) ** -0.5).to(DEV)
    bias = (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
    ids = torch.randint(0, M, (B, L), generator=gen, dtype=torch.int32).to(DEV)
    row_max = torch.ones(B, M, dtype=torch.int32, device=DEV)
    row_max[:, 1 % M] = 0
    with ops.precision("bf16"):
        wp = ops.pointmlp_pack(W, "bf16")
        one = ops.const_vec(Cout, 1.0, DEV)
        if not ops.pointmlp_bf16_pool_ok(r1, r2, Cout, M):
            pytest.skip("shape outside the pooled launch")
        y1 = ops.channel_affine_act(r1, s1, h1, True)
        y2 = ops.channel_affine_act(r2, s2, h2, True) if C2 else None
        ref = ops.pointmlp_bf16_pool(y1, wp, one, bias, False, Cout, ids, M, row_max, x2=y2)
        got = ops.pointmlp_bf16_pool(r1, wp, one, bias, False, Cout, ids, M, row_max, x2=r2,
                                     xaff=(s1, h1, True) + ((s2, h2, True) if C2 else ()))
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


@pytest.mark.parametrize("B,Cout,Cin,L", [(64, 128, 64, 15000), (64, 256, 128, 15000), (40, 70, 45, 8008)])
def test_bf16_wgrad_with_normalise_on_load_equals_wgrad_on_the_normalised_tensor(B, Cout, Cin, L):
    from sonet_hip import ops
    raw, sc, sh = _raw_case(B, Cin, L, Cout + L)
    g = (torch.randn(B, Cout, L, generator=torch.Generator().manual_seed(L)) * 1e-3).to(torch.bfloat16).to(DEV)
    assert ops.wgrad_bf16_xaff_ok(B, Cout, Cin, L)
    for relu in (True, False):
        y = ops.channel_affine_act(raw, sc, sh, relu)
        assert torch.equal(ops.wgrad_bf16(g, raw, xaff=(sc, sh, relu)), ops.wgrad_bf16(g, y)), relu


@pytest.mark.parametrize("B,C,M,Ci,L", [(64, 384, 64, 64, 15000), (3, 384, 64, 256, 3000), (2, 96, 8, 47, 131)])
def test_bf16_pooled_wgrad_with_normalise_on_load_equals_pooled_wgrad_on_the_normalised_tensor(B, C, M, Ci, L):
    from sonet_hip import ops
    raw, sc, sh = _raw_case(B, Ci, L, C + L)
    gen = torch.Generator().manual_seed(B + L)
    g_t = torch.randn(B, M, C, generator=gen).to(DEV)
    pos_t = torch.randint(0, L, (B, M, C), generator=gen, dtype=torch.int32).to(DEV)
    for relu in (True, False):
        y = ops.channel_affine_act(raw, sc, sh, relu)
        assert torch.equal(ops.pooled_wgrad(g_t, pos_t, raw, (sc, sh, relu)), ops.pooled_wgrad(g_t, pos_t, y)), relu


def test_bf16_training_step_with_normalise_on_load_is_bit_identical():
    """ops.BF16_NORM_ON_LOAD on / off: same kernels' arithmetic on the same values -- the loss, every gradient, the running statistics and the
    pooled map of three consecutive steps (Adam updates in between) are bit-identical, and no normalise pass runs on a point-level tensor."""
    from argparse import Namespace
    from models import networks as NW
    from sonet_hip import ops, synth
    from sonet_hip.optim import FusedAdam
    B, N = 64, 5000                                                # (the streaming kernels' shapes: >= 8192 column groups)
    out = {}
    old = ops.BF16_NORM_ON_LOAD
    try:
        with ops.precision("bf16"):
            for flag in (True, False):
                ops.BF16_NORM_ON_LOAD = flag
                opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                                activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg", bn_momentum=0.1,
                                bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                oe, oc = FusedAdam(enc.parameters(), lr=1e-3), FusedAdam(cls.parameters(), lr=1e-3)
                inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
                snaps = []
                for it in range(3):
                    with ops.kernel_timing() as rec:
                        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                        enc.zero_grad(set_to_none=True)
                        cls.zero_grad(set_to_none=True)
                        loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                        loss.backward()
                    names = [n for n, _, _ in rec.records]
                    big_norm = sum(1 for n, _, _ in rec.records if n == "channel_affine_act_bf16")
                    if flag:
                        assert any(n.endswith("_xaff") and n.startswith("pointmlpbf16_pool") for n in names), names
                        assert any(n.endswith("_xaff") and n.startswith("wgradbf16") for n in names) and "pooled_wgrad_xaff" in names
                    snaps.append((loss.detach().clone(), {k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(cls.named_parameters()) if p.grad is not None},
                                  {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone(), big_norm))
                    oe.step()
                    oc.step()
                out[flag] = snaps
    finally:
        ops.BF16_NORM_ON_LOAD = old
    for it in range(3):
        a, b = out[True][it], out[False][it]
        assert a[4] == b[4] - 3, (a[4], b[4])                      # the three point-level normalise passes are gone, nothing else
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
        assert a[1].keys() == b[1].keys()
        for k in a[1]:
            assert torch.equal(a[1][k], b[1][k]), (it, k)
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), (it, k)


# ---- BatchNorm / ReLU backward on the operand load of the bf16 input-gradient launch (sonet_pointmlp_bf16_bnb) -----------------------------

def _bnb_case(B, C, L, seed):
    g = torch.Generator().manual_seed(seed)
    gy = (torch.randn(B, C, L, generator=g) * 1e-3).to(torch.bfloat16).to(DEV)
    raw = (torch.randn(B, C, L, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    co = [(torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV),
          (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.4).to(DEV)]
    return gy, raw, co


@pytest.mark.parametrize("B,C,Cin,L", [(64, 256, 128, 15000), (64, 128, 64, 15000), (5, 64, 64, 1002), (3, 512, 192, 4098), (2, 32, 64, 66)])
@pytest.mark.parametrize("acc", [False, True])
def test_bf16_dgrad_with_batchnorm_backward_on_load_equals_apply_pass_then_dgrad(B, C, Cin, L, acc):
    """sonet_pointmlp_bf16_bnb == sonet_pointwise_bwd_apply_bf16 followed by sonet_pointmlp_bf16 (or _acc) on the transposed pack: the input
    gradient AND the g_raw it stores for the weight gradient, bit for bit; with and without the ReLU mask, with and without the g_raw store."""
    from sonet_hip import ops
    gy, raw, (a, b, c0, sc, sh) = _bnb_case(B, C, L, C + L)
    gen = torch.Generator().manual_seed(7 + Cin)
    W = (torch.randn(C, Cin, generator=gen) * C ** -0.5).to(DEV)       # the layer's weight: C outputs, Cin inputs; the dgrad multiplies by W^T
    addend = (torch.randn(B, Cin, L, generator=gen) * 1e-3).to(torch.bfloat16).to(DEV) if acc else None
    assert ops.pointmlp_bf16_bnb_ok(C, Cin, L)
    with ops.precision("bf16"):
        wpt = ops.pointmlp_pack_transposed(W, 0, Cin, Cin, "bf16")
        one, zero = ops.const_vec(Cin, 1.0, DEV), ops.const_vec(Cin, 0.0, DEV)
        for relu in (True, False):
            g_ref = ops.pointwise_bwd_apply(gy, raw, sc, sh, relu, a, b, c0)
            y_ref = ops.pointmlp(g_ref, wpt, one, zero, False, Cin, acc=addend)
            y, g = ops.pointmlp_bf16_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cin, want_g_raw=True, acc=addend)
            assert y.dtype == torch.bfloat16 and g.dtype == torch.bfloat16
            assert torch.equal(g.view(torch.int16), g_ref.view(torch.int16)), relu
            assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), relu
            y2, g2 = ops.pointmlp_bf16_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cin, want_g_raw=False, acc=addend)
            assert g2 is None and torch.equal(y2.view(torch.int16), y_ref.view(torch.int16))


def test_bf16_bnb_rejects_shapes_it_does_not_take():
    from sonet_hip import ops
    from sonet_hip._lib import SonetHipError
    gy, raw, (a, b, c0, sc, sh) = _bnb_case(2, 48, 65, 3)             # odd L
    W = torch.randn(48, 64, device=DEV)
    with ops.precision("bf16"):
        wpt = ops.pointmlp_pack_transposed(W, 0, 64, 64, "bf16")
        with pytest.raises(SonetHipError):
            ops.pointmlp_bf16_bnb(gy, raw, wpt, ops.const_vec(64, 1.0, DEV), ops.const_vec(64, 0.0, DEV), a, b, c0, sc, sh, True, 64)


def test_bf16_training_step_with_batchnorm_backward_on_load_is_bit_identical():
    """ops.BF16_BNB_ON_LOAD on / off: the loss, every gradient and the running statistics of three consecutive steps are bit-identical, the
    apply pass of the two hidden layers that have an input gradient is gone (the first layer's stays: it has no input gradient to ride on)."""
    from argparse import Namespace
    from models import networks as NW
    from sonet_hip import ops, synth
    from sonet_hip.optim import FusedAdam
    B, N = 64, 5000
    out = {}
    old = ops.BF16_BNB_ON_LOAD
    try:
        with ops.precision("bf16"):
            for flag in (True, False):
                ops.BF16_BNB_ON_LOAD = flag
                opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                                activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg", bn_momentum=0.1,
                                bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                oe, oc = FusedAdam(enc.parameters(), lr=1e-3), FusedAdam(cls.parameters(), lr=1e-3)
                inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
                snaps = []
                for it in range(3):
                    with ops.kernel_timing() as rec:
                        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                        enc.zero_grad(set_to_none=True)
                        cls.zero_grad(set_to_none=True)
                        loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                        loss.backward()
                    names = [n for n, _, _ in rec.records]
                    n_bnb = sum(1 for n in names if n.startswith("pointmlpbf16_bnb"))
                    assert n_bnb == (2 if flag else 0), names
                    snaps.append((loss.detach().clone(), {k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(cls.named_parameters()) if p.grad is not None},
                                  {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, names))
                    oe.step()
                    oc.step()
                out[flag] = snaps
    finally:
        ops.BF16_BNB_ON_LOAD = old
    for it in range(3):
        a, b = out[True][it], out[False][it]
        assert torch.equal(a[0], b[0])
        assert a[1].keys() == b[1].keys()
        for k in a[1]:
            assert torch.equal(a[1][k], b[1][k]), (it, k)
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), (it, k)
        big = lambda names: sum(1 for n in names if n.startswith("pointwise_bwd_apply"))      # noqa: E731
        assert big(a[3]) == big(b[3]) - 2, (big(a[3]), big(b[3]))
