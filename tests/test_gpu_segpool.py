"""GPU tests of the f32-class training forward's pooled last layer on NODE-SORTED columns (sonet_pointmlp_h3_segpool_f32:
models/layers.py:431 + models/networks.py:180-185 with models/index_max_ext/index_max_cuda.cu:10-26 as the pool's definition).

Positions: bit-exact.  Values: bit-exact against the storing launch of the same arithmetic (+0 for -0)."""
from argparse import Namespace

import pytest
import torch

from conftest import assert_close_rms

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _sorted_case(B, C1, C2, Cout, L, M, seed):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(B, C1, L, generator=g).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(DEV) if C2 else None
    if L >= 8:
        x1[:, :, 5] = x1[:, :, 3]                                 # exact ties between columns 3 and 5
        if x2 is not None:
            x2[:, :, 5] = x2[:, :, 3]
    W = (torch.randn(Cout, C1 + C2, generator=g) * (C1 + C2) ** -0.5).to(DEV)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    bias[1 % Cout] = -3000.0                                      # a channel that never beats -1000
    ids = torch.randint(0, M, (B, L), generator=g, dtype=torch.int32)
    if M > 3:
        ids[ids == 2] = 3                                          # node 2 is empty
    if L >= 8:
        ids[B - 1, 0] = -1                                         # ids nobody owns (sorted: one below, one above the range)
        ids[B - 1, 1] = M + 5
    ids = torch.sort(ids, dim=1).values.contiguous()
    if L >= 8:
        ids[0, 5] = ids[0, 3]
        ids[0, 4] = ids[0, 3]                                      # (keeps the row sorted: columns 3..5 in one node -> the tie is inside a node)
    pos0 = torch.randint(0, L, (B,), generator=g, dtype=torch.int32)
    row_max = torch.ones(B, M, dtype=torch.int32)
    row_max[:, 1 % M] = 0                                          # a masked node
    return x1, x2, W, bias, ids.to(DEV), pos0.to(DEV), row_max.to(DEV)


@pytest.mark.parametrize("B,C1,C2,Cout,L,M", [(64, 64, 256, 384, 15000, 64), (3, 64, 256, 384, 1000, 64), (2, 128, 0, 96, 130, 7),
                                               (5, 64, 0, 64, 2, 3), (2, 64, 64, 128, 4098, 255), (2, 16, 0, 32, 33, 1)])
def test_h3_layer_with_sorted_pool_epilogue_equals_layer_then_index_max(B, C1, C2, Cout, L, M):
    """sonet_pointmlp_h3_segpool_f32 == sonet_pointmlp_h3_f32 followed by sonet_index_max_gather_f32 on the stored tensor: positions and
    values bit for bit wherever a column beat -1000; bins nothing beat (empty nodes, the -3000 channel), masked nodes: pos0[b] and the
    layer's value there; ids outside [0, M) ignored; ties: the first column wins."""
    from sonet_hip import ops
    x1, x2, W, bias, ids, pos0, row_max = _sorted_case(B, C1, C2, Cout, L, M, B + C1 + Cout + L + M)
    with ops.precision("h3"):
        wp = ops.pointmlp_pack(W, "h3")
        one = ops.const_vec(Cout, 1.0, DEV)
        assert ops.pointmlp_h3_segpool_ok(x1, x2, wp, Cout, M)
        y = ops.pointmlp(x1, wp, one, bias, False, Cout, x2=x2)
        idx_ref, val_ref = ops.index_max_gather(y, ids, M, None)
        # which bins did something beat -1000 in?
        valid = (ids >= 0) & (ids < M)
        tgt = torch.where(valid, ids, torch.full_like(ids, M)).long().unsqueeze(1).expand(B, Cout, L)
        mx = torch.full((B, Cout, M + 1), float("-inf"), device=DEV).scatter_reduce(2, tgt, y, "amax")[:, :, :M]
        won0 = mx > -1000.0
        at0 = torch.gather(y, 2, pos0.long().view(B, 1, 1).expand(B, Cout, M))
        for rm in (row_max, None):
            idx, val = ops.pointmlp_h3_segpool(x1, wp, one, bias, False, Cout, ids, pos0, M, rm, x2=x2)
            won = won0 if rm is None else won0 & (rm.unsqueeze(1) != 0)
            exp_idx = torch.where(won, idx_ref, pos0.view(B, 1, 1).expand(B, Cout, M))
            exp_val = torch.where(won, val_ref, at0)
            assert torch.equal(idx, exp_idx)
            assert torch.equal(val, exp_val)                      # (== : -0 and +0 compare equal)
            assert torch.equal(torch.gather(y, 2, idx.long()), val)
        idx2, val2 = ops.pointmlp_h3_segpool(x1, wp, one, bias, False, Cout, ids, pos0, M, None, x2=x2)
        assert torch.equal(idx2, idx) and torch.equal(val2, val)  # deterministic (integer max of keys)
    assert bool((~won0[:, 1 % Cout]).all())                       # the -3000 channel: nothing beat -1000


def _rel_rms(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).square().mean().sqrt() / b.square().mean().sqrt().clamp_min(1e-30))


def _compare_steps(got, ref):
    """(loss, grads, running statistics, pooled map) of two training steps of the same arithmetic class whose BatchNorm sums ran in
    different orders.  Forward: f32-class.  Gradients: the network has three arg-max pools (per node, over the K neighbours, over the
    nodes) and every bin routes its gradient to ONE winner; a 1e-7 change of the forward flips a few of the 10^5 winners, and one flip
    moves a weight gradient by ~sqrt(2 / bins) rel-rms (tools/grad_dev_h3.py, profiles/r05c_grad_dev_h3.log: the reference's own float32
    run sits 1e-3 .. 5e-3 from its float64 run for the same reason) -- bounded at that level here; the kernels are pinned bit for bit
    one by one above."""
    assert abs(float(got[0]) - float(ref[0])) <= 1e-5 * abs(float(ref[0]))
    assert_close_rms(got[3].cpu().numpy(), ref[3].cpu().numpy(), 1e-5, "first_pn_out_masked_max")
    assert got[1].keys() == ref[1].keys()
    for k in got[1]:
        # (conv biases sit in front of a training-mode BatchNorm -- directly, or through the pools for the first PointNet's last layer --:
        #  true gradient zero, both sides hold rounding noise)
        if k.endswith("conv.bias"):
            continue
        assert _rel_rms(got[1][k], ref[1][k]) <= 1e-2, (k, _rel_rms(got[1][k], ref[1][k]))
    for k in got[2]:
        assert_close_rms(got[2][k].cpu().numpy(), ref[2][k].cpu().numpy(), 1e-5, k)


def _opt(B, N):
    return Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024, activation="relu",
                     normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg", bn_momentum=0.1,
                     bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)


def test_h3_training_step_on_sorted_columns_equals_the_storing_path():
    """The f32-class training step with the first PointNet on node-sorted columns and the pool in the last layer's epilogue == the step in
    the original column order that stores first_pn_out and runs index_max on it: the pooled map, the loss, every gradient and the updated
    BatchNorm statistics agree to f32-class accuracy (the sums run in another order: not bit for bit)."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 4, 1500
    res = {}
    with ops.precision("h3"):
        for flag in (True, False):
            old = ops.H3_SEGPOOL
            ops.H3_SEGPOOL = flag
            try:
                opt = _opt(B, N)
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
                assert any(n.startswith("pointmlph3_segpool") for n in names) == flag
                assert any(n.startswith("index_max") for n in names) == (not flag)
                if flag:
                    with pytest.raises(RuntimeError, match="want_first_pn_out"):
                        enc.first_pn_out
                    assert tuple(enc.x_decentered.shape) == (B, 3, 3 * N) and tuple(enc.centers.shape) == (B, 3, 3 * N)   # (lazy attributes)
                res[flag] = (loss.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                             {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone(),
                             enc.som_node.clone())
            finally:
                ops.H3_SEGPOOL = old
    assert_close_rms(res[True][4].cpu().numpy(), res[False][4].cpu().numpy(), 1e-6, "som_node")   # (f64 sums in another order)
    _compare_steps(res[True], res[False])


def test_sorted_pool_handles_an_empty_node_like_the_reference():
    """A cloud with an EMPTY node: the reference gathers position 0 of first_pn_out for it (models/networks.py:185) and its gradient flows to
    original column 0.  On sorted columns that column sits at pos0[b]: same pooled map, same gradients as the storing path."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 2, 700
    grads = {}
    with ops.precision("h3"):
        for flag in (True, False):
            old = ops.H3_SEGPOOL
            ops.H3_SEGPOOL = flag
            try:
                opt = _opt(B, N)
                enc = NW.Encoder(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 5)
                enc.to(DEV).train()
                inp = synth.make_inputs(B, N, seed=21, device=torch.device(DEV))
                node = inp["node"].clone()
                node[:, :, 7] = 50.0                               # a node far away from every point: empty
                feat = enc(inp["pc"], inp["sn"], node, inp["node_knn_I"], True, 0)
                assert int(enc._lazy["a"].count[:, 7].sum()) == 0     # (nobody chose node 7)
                mm = enc.first_pn_out_masked_max
                (feat.square().mean() + mm[:, :, 7].square().mean()).backward()
                grads[flag] = ({k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None}, mm.detach().clone())
            finally:
                ops.H3_SEGPOOL = old
    assert_close_rms(grads[True][1].cpu().numpy(), grads[False][1].cpu().numpy(), 1e-5, "masked max")
    for k in grads[True][0]:
        if k.startswith("first_pointnet") and not k.endswith("conv.bias"):
            assert _rel_rms(grads[True][0][k], grads[False][0][k]) <= 1e-2, k          # (winner flips: see _compare_steps)


# ------------------------------------------------------------------------------------------ normalise-on-load
def _raw_case(B, C, L, seed):
    g = torch.Generator().manual_seed(seed)
    raw = (torch.randn(B, C, L, generator=g) * 1.7 + 0.3).to(DEV)
    sc = (torch.rand(C, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(C, generator=g) * 0.4).to(DEV)
    return raw, sc, sh


@pytest.mark.parametrize("B,C1,C2,Cout,L", [(64, 64, 0, 128, 15000), (16, 128, 0, 256, 4100), (16, 64, 256, 384, 4100), (16, 48, 0, 128, 4099)])
def test_layer_with_normalise_on_load_equals_layer_on_the_normalised_tensor(B, C1, C2, Cout, L):
    """sonet_pointmlp_h3_stats_xaff_f32 on RAW inputs == sonet_pointmlp_h3_stats_f32 on sonet_channel_affine_act_f32's output: the
    operand load does the same fma + ReLU, so the outputs agree bit for bit (and the batch statistics to summation order)."""
    from sonet_hip import ops
    r1, s1, h1 = _raw_case(B, C1, L, 1 + C1 + L)
    r2, s2, h2 = _raw_case(B, C2, L, 2 + C2 + L) if C2 else (None, None, None)
    gen = torch.Generator().manual_seed(Cout)
    W = (torch.randn(Cout, C1 + C2, generator=gen) * (C1 + C2) ** -0.5).to(DEV)
    bias = (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
    with ops.precision("h3"):
        wp = ops.pointmlp_pack(W, "h3")
        one = ops.const_vec(Cout, 1.0, DEV)
        assert ops.xaff_ok(C1, C2, Cout)
        y1 = ops.channel_affine_act(r1, s1, h1, True)
        y2 = ops.channel_affine_act(r2, s2, h2, False) if C2 else None
        ref = ops.pointmlp_stats(y1, wp, one, bias, False, Cout, x2=y2)
        xa = (s1, h1, True) + ((s2, h2, False) if C2 else ())
        got = ops.pointmlp_stats(r1, wp, one, bias, False, Cout, x2=r2, xaff=xa)
    assert torch.equal(got[0], ref[0])
    for a, b, n in zip(got[1:], ref[1:], ("mean", "var")):        # (the reference launch may be the second-generation kernel: other partial sums)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), n


@pytest.mark.parametrize("B,Cout,Cin,L", [(64, 128, 64, 15000), (8, 256, 128, 4100), (3, 70, 45, 1001)])
def test_wgrad_with_normalise_on_load_equals_wgrad_on_the_normalised_tensor(B, Cout, Cin, L):
    from sonet_hip import ops
    raw, sc, sh = _raw_case(B, Cin, L, Cout + L)
    g = (torch.randn(B, Cout, L, generator=torch.Generator().manual_seed(L)) * 1e-3).to(DEV)
    for relu in (True, False):
        y = ops.channel_affine_act(raw, sc, sh, relu)
        assert torch.equal(ops.wgrad_x3(g, raw, xaff=(sc, sh, relu)), ops.wgrad_x3(g, y)), relu


@pytest.mark.parametrize("B,C,M,Ci,L", [(64, 384, 64, 64, 15000), (3, 384, 64, 256, 3000), (2, 96, 8, 48, 131)])
def test_pooled_wgrad_with_normalise_on_load_equals_pooled_wgrad_on_the_normalised_tensor(B, C, M, Ci, L):
    from sonet_hip import ops
    raw, sc, sh = _raw_case(B, Ci, L, C + L)
    gen = torch.Generator().manual_seed(B + L)
    g_t = torch.randn(B, M, C, generator=gen).to(DEV)
    pos_t = torch.randint(0, L, (B, M, C), generator=gen, dtype=torch.int32).to(DEV)
    y = ops.channel_affine_act(raw, sc, sh, True)
    assert torch.equal(ops.pooled_wgrad(g_t, pos_t, raw, (sc, sh, True)), ops.pooled_wgrad(g_t, pos_t, y))


def test_sorted_pool_with_normalise_on_load_equals_sorted_pool_on_the_normalised_tensors():
    from sonet_hip import ops
    B, C1, C2, Cout, L, M = 8, 64, 256, 384, 5001, 64
    _, _, W, bias, ids, pos0, row_max = _sorted_case(B, C1, C2, Cout, L, M, 77)
    r1, s1, h1 = _raw_case(B, C1, L, 5)
    r2, s2, h2 = _raw_case(B, C2, L, 6)
    with ops.precision("h3"):
        wp = ops.pointmlp_pack(W, "h3")
        one = ops.const_vec(Cout, 1.0, DEV)
        y1, y2 = ops.channel_affine_act(r1, s1, h1, True), ops.channel_affine_act(r2, s2, h2, True)
        ref = ops.pointmlp_h3_segpool(y1, wp, one, bias, False, Cout, ids, pos0, M, row_max, x2=y2)
        got = ops.pointmlp_h3_segpool(r1, wp, one, bias, False, Cout, ids, pos0, M, row_max, x2=r2, xaff=(s1, h1, True, s2, h2, True))
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_h3_training_step_with_normalise_on_load_equals_the_step_with_normalise_passes():
    """Same sorted columns, same kernels' arithmetic: only the 128 x 64 weight gradient changes kernels (hipBLASLt -> the split-operand kernel
    with normalise-on-load), so everything agrees to f32-class accuracy -- and no normalise pass runs on the point-level tensors."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 8, 3000                                                 # (big enough for the statistics epilogue: B x kN x 128 x 4 >= 32 MB)
    res = {}
    with ops.precision("h3"):
        for flag in (True, False):
            old = ops.H3_NORM_ON_LOAD
            ops.H3_NORM_ON_LOAD = flag
            try:
                opt = _opt(B, N)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
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
                assert any(n.endswith("_xaff") for n in names) == flag
                assert any(n.startswith("pointmlph3_segpool") for n in names)
                res[flag] = (loss.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                             {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone())
            finally:
                ops.H3_NORM_ON_LOAD = old
    _compare_steps(res[True], res[False])


@pytest.mark.parametrize("which", [1, 2])
def test_h3_training_step_with_a_hidden_layer_that_materialises_its_input(which, monkeypatch):
    """A hidden layer of the first PointNet that finds no normalise-on-load form for itself AFTER the forward decided on deferred
    activations (the weight side of the range guard can do that between the decision and the launch) writes its input out through the
    identity node ``_Materialise`` and carries on; layer 1 doing so is also the consumer of the carried gradient (its input-gradient launch
    then has another input tensor than the pooled layer).  Same step as the plain one to f32-class accuracy."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 8, 3000
    res = {}
    real = ops.xaff_ok
    with ops.precision("h3"):
        for patched in (True, False):
            opt = _opt(B, N)
            enc, cls = NW.Encoder(opt), NW.Classifier(opt)
            enc.want_first_pn_out = False
            synth.fill_state_dict_(enc.state_dict(), 3)
            synth.fill_state_dict_(cls.state_dict(), 4)
            enc.to(DEV).train()
            cls.to(DEV).train()
            inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
            if patched:
                lay = enc.first_pointnet.layers[which]
                run = lay._run

                def run_without_the_form(*a, _run=run, **k):
                    monkeypatch.setattr(ops, "xaff_ok", lambda *aa, **kk: False)
                    try:
                        return _run(*a, **k)
                    finally:
                        monkeypatch.setattr(ops, "xaff_ok", real)
                object.__setattr__(lay, "_run", run_without_the_form)
            with ops.kernel_timing() as rec:
                feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                loss.backward()
            names = [n for n, _, _ in rec.records]
            assert any(n.startswith("pointmlph3_segpool") for n in names)
            # (the materialised input is one normalise pass more on a point-level tensor)
            res[patched] = (loss.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                            {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone(),
                            sum(n.startswith("channel_affine_act") for n in names))
    assert res[True][4] == res[False][4] + 1
    _compare_steps(res[True][:4], res[False][:4])


# ------------------------------------------------------------------------------------------ deferred side-stream joins
@pytest.mark.parametrize("precision", ["bf16", "h3"])
def test_weight_gradients_joined_at_the_end_of_backward_equal_joined_per_layer(precision):
    """``ops.DEFER_WGRAD_JOIN``: the side stream that computes the weight gradients is joined at the end of the backward pass (an engine
    callback; the gradient all-reducer's hooks join earlier) instead of at the end of every layer's backward.  Same kernels, same inputs:
    on the bit-reproducible paths (bf16; f32-class in the original column order) every gradient of three consecutive steps -- with an Adam
    update in between, memory released and re-used -- must be bit-identical to the per-layer joins.  (A gradient read before its side-stream
    launch has finished, or an operand whose memory was handed out again, would show here.)"""
    from models import networks as NW
    from sonet_hip import ops, synth
    from sonet_hip.optim import FusedAdam
    B, N = 6, 2200
    out = {}
    old = (ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL)
    try:
        ops.H3_SEGPOOL = False                                     # (the sorted path's BatchNorm sums are not run-to-run reproducible)
        with ops.precision(precision):
            for flag in (True, False):
                ops.DEFER_WGRAD_JOIN = flag
                opt = _opt(B, N)
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
                    feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                    enc.zero_grad(set_to_none=True)
                    cls.zero_grad(set_to_none=True)
                    loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                    loss.backward()
                    junk = [torch.empty(1 << 22, device=DEV).fill_(float(it)) for _ in range(8)]      # memory churn right behind the backward
                    del junk
                    snaps.append({k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(cls.named_parameters()) if p.grad is not None})
                    oe.step()
                    oc.step()
                out[flag] = snaps
    finally:
        ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL = old
    for it in range(3):
        assert out[True][it].keys() == out[False][it].keys()
        for k in out[True][it]:
            assert torch.equal(out[True][it][k], out[False][it][k]), (it, k)


def test_shared_weights_and_hooked_weights_join_their_side_streams_in_time():
    """a weight with TWO producers in one backward pass (the encoder called twice before one backward: a Siamese use) has
    its two gradients summed by the autograd engine on the main stream as soon as both exist, and a tensor hook on a weight reads its
    gradient inside the pass -- neither may see a side-stream launch that is still running.  The deferred join is therefore taken only
    for single-producer, hook-free weights (models/layers.py ``_grad_slot_empty``); here: the gradients of a two-call step and what a
    hook saw are bit-identical to per-layer joins (ops.DEFER_WGRAD_JOIN = False)."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 6, 2200
    out, seen = {}, {}
    old = (ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL)
    try:
        ops.H3_SEGPOOL = False                                     # (bit-reproducible path)
        with ops.precision("h3"):
            for flag in (True, False):
                ops.DEFER_WGRAD_JOIN = flag
                opt = _opt(B, N)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                a, b = synth.make_inputs(B, N, seed=9, device=torch.device(DEV)), synth.make_inputs(B, N, seed=10, device=torch.device(DEV))
                hooked = enc.first_pointnet.layers[1].conv.weight
                seen[flag] = []
                h = hooked.register_hook(lambda g_, log=seen[flag]: log.append(g_.clone()))
                fa = enc(a["pc"], a["sn"], a["node"], a["node_knn_I"], is_train=True, epoch=0)
                fb = enc(b["pc"], b["sn"], b["node"], b["node_knn_I"], is_train=True, epoch=0)
                loss = torch.nn.functional.cross_entropy(cls(fa, 0), a["label"]) + torch.nn.functional.cross_entropy(cls(fb, 0), b["label"])
                loss.backward()
                junk = [torch.empty(1 << 22, device=DEV).fill_(1.0) for _ in range(8)]
                del junk
                h.remove()
                out[flag] = {k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(cls.named_parameters()) if p.grad is not None}
    finally:
        ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL = old
    assert out[True].keys() == out[False].keys() and len(out[True]) > 30
    for k in out[True]:
        assert torch.equal(out[True][k], out[False][k]), k
    assert len(seen[True]) == len(seen[False]) >= 1
    for x, y in zip(seen[True], seen[False]):
        assert torch.equal(x, y)


def test_deferred_join_survives_a_training_loop_that_keeps_the_previous_loss():
    """``loss = step()`` keeps the previous step's loss -- its graph nodes and their use tokens -- alive while the next step runs.  Those uses ran
    in an EARLIER pass and must not switch the deferral off (round 6: they did, for every step but the first); the gradients of every step
    are bit-identical to per-layer joins."""
    from models import layers as L
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 6, 2200
    out, counts = {}, {}
    real = L._grad_slot_empty
    old = (ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL)
    try:
        ops.H3_SEGPOOL = False                                     # (bit-reproducible path)
        with ops.precision("h3"):
            for flag in (True, False):
                ops.DEFER_WGRAD_JOIN = flag
                verdicts = []
                L._grad_slot_empty = lambda w, t=None, v=verdicts: (v.append(real(w, t)) or v[-1])
                opt = _opt(B, N)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                a = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
                torch.manual_seed(3)
                loss, grads, per_step = None, [], []
                for it in range(3):
                    del verdicts[:]
                    feat = enc(a["pc"], a["sn"], a["node"], a["node_knn_I"], is_train=True, epoch=0)
                    enc.zero_grad(set_to_none=True)
                    cls.zero_grad(set_to_none=True)
                    new_loss = torch.nn.functional.cross_entropy(cls(feat, 0), a["label"])
                    new_loss.backward()                            # (the previous step's ``loss`` is still alive here)
                    loss = new_loss
                    per_step.append(sum(verdicts))
                    grads.append({k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(cls.named_parameters()) if p.grad is not None})
                out[flag], counts[flag] = grads, per_step
    finally:
        L._grad_slot_empty = real
        ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL = old
    assert counts[True][0] >= 4 and counts[True][1] == counts[True][0] and counts[True][2] == counts[True][0], counts
    for it in range(3):
        assert out[True][it].keys() == out[False][it].keys()
        for k in out[True][it]:
            assert torch.equal(out[True][it][k], out[False][it][k]), (it, k)


# ------------------------------------------------------------------------------------------ BatchNorm backward on load
@pytest.mark.parametrize("B,C,Cout,L,relu", [(64, 256, 128, 15000, True), (8, 128, 64, 4100, True), (3, 512, 512, 577, False), (2, 48, 96, 131, True)])
def test_dgrad_with_batchnorm_backward_on_load_equals_apply_then_dgrad(B, C, Cout, L, relu):
    """sonet_pointmlp_x3_bnb_f32 == sonet_pointwise_bwd_apply_f32 followed by sonet_pointmlp_x3_f32 on its output: the input gradient and
    the g_raw side output, bit for bit (incl. channel counts that are not multiples of 16 and ragged column tails)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C + L)
    gy = (torch.randn(B, C, L, generator=g) * 1e-3).to(DEV)
    raw = (torch.randn(B, C, L, generator=g) * 1.5).to(DEV)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV)
    a, b, c0 = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV), (torch.randn(C, generator=g) * 1e-5).to(DEV)
    Wt = (torch.randn(Cout, C, generator=g) * C ** -0.5).to(DEV)           # the dgrad's matrix: Cout (= the layer's Cin) x C
    wpt = ops.pointmlp_pack(Wt, "x3")
    one, zero = ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV)
    g_raw_ref = ops.pointwise_bwd_apply(gy, raw, sc, sh, relu, a, b, c0)
    y_ref = ops.pointmlp(g_raw_ref, wpt, one, zero, False, Cout)
    y, g_raw = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout)
    assert torch.equal(g_raw, g_raw_ref)
    assert torch.equal(y, y_ref)
    y2, none = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout, want_g_raw=False)
    assert none is None and torch.equal(y2, y_ref)


@pytest.mark.parametrize("B,C,Cout,L,relu", [(64, 128, 64, 15000, True), (3, 256, 128, 577, True), (2, 48, 96, 131, False)])
def test_dgrad_with_accumulating_store_equals_dgrad_plus_the_other_gradient(B, C, Cout, L, relu):
    """sonet_pointmlp_x3_bnb_acc_f32 == sonet_pointmlp_x3_bnb_f32 followed by the f32 addition autograd's accumulation would run, bit for
    bit; the g_raw side output is untouched by the addend."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C + L + 1)
    gy = (torch.randn(B, C, L, generator=g) * 1e-3).to(DEV)
    raw = (torch.randn(B, C, L, generator=g) * 1.5).to(DEV)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV)
    a, b, c0 = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV), (torch.randn(C, generator=g) * 1e-5).to(DEV)
    Wt = (torch.randn(Cout, C, generator=g) * C ** -0.5).to(DEV)
    other = (torch.randn(B, Cout, L, generator=g) * 1e-3).to(DEV)
    other[:, :, ::7] = 0.0                                                 # (the sparse gradient of the pooled layer: mostly zeros)
    wpt = ops.pointmlp_pack(Wt, "x3")
    one, zero = ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV)
    y_ref, g_raw_ref = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout)
    keep = other.clone()
    y, g_raw = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout, acc=other)
    assert torch.equal(other, keep)
    assert torch.equal(g_raw, g_raw_ref)
    assert torch.equal(y, y_ref + other) and torch.equal(y, other + y_ref)
    with pytest.raises(ops.SonetHipError):
        ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cout, acc=other[:, :, :-1].contiguous())


def test_training_step_with_the_carried_gradient_is_bit_identical():
    """ops.GRAD_CARRY on / off on the node-sorted f32-class path (bit-reproducible: deterministic sort): the gradient the last layer of the
    first PointNet sends to the first layer's output is added by the second layer's input-gradient launch instead of by autograd --
    every gradient identical; also with a second backward pass through a retained graph (a deposit is consumed once per pass)."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 8, 3000
    out = {}
    old = ops.GRAD_CARRY
    try:
        with ops.precision("h3"):
            for flag in (True, False):
                ops.GRAD_CARRY = flag
                opt = _opt(B, N)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
                with ops.kernel_timing() as rec:
                    feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                    loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                    loss.backward(retain_graph=True)
                names = [n for n, _, _ in rec.records]
                assert any(n.startswith("pointmlpx3_bnba_") for n in names) == flag, names
                out[flag] = {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}
                enc.zero_grad(set_to_none=True)
                cls.zero_grad(set_to_none=True)
                loss.backward()
                again = {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}
                for k in again:
                    assert torch.equal(again[k], out[flag][k]), ("second pass", flag, k)
    finally:
        ops.GRAD_CARRY = old
    assert out[True].keys() == out[False].keys()
    for k in out[True]:
        assert torch.equal(out[True][k], out[False][k]), k


def test_training_step_with_batchnorm_backward_on_load_is_bit_identical():
    """ops.BNB_ON_LOAD on / off on the bit-reproducible f32-class path (original column order): every gradient identical."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 6, 2200
    out = {}
    old = (ops.BNB_ON_LOAD, ops.H3_SEGPOOL)
    try:
        ops.H3_SEGPOOL = False
        with ops.precision("h3"):
            for flag in (True, False):
                ops.BNB_ON_LOAD = flag
                opt = _opt(B, N)
                enc, cls = NW.Encoder(opt), NW.Classifier(opt)
                enc.want_first_pn_out = False
                synth.fill_state_dict_(enc.state_dict(), 3)
                synth.fill_state_dict_(cls.state_dict(), 4)
                enc.to(DEV).train()
                cls.to(DEV).train()
                inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
                with ops.kernel_timing() as rec:
                    feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                    torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"]).backward()
                assert any(n.startswith("pointmlpx3_bnb") for n, _, _ in rec.records) == flag
                out[flag] = {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}
    finally:
        ops.BNB_ON_LOAD, ops.H3_SEGPOOL = old
    assert out[True].keys() == out[False].keys()
    for k in out[True]:
        assert torch.equal(out[True][k], out[False][k]), k


# ------------------------------------------------------------------------------------------ deterministic sort of the training forward
@pytest.mark.parametrize("B,N,M,k", [(64, 5000, 64, 3), (3, 700, 64, 3), (2, 513, 128, 2), (2, 100, 7, 1)])
def test_deterministic_sort_is_stable_and_equal_to_the_default_up_to_the_order_inside_a_node(B, N, M, k):
    """sonet_som_assign_sort_det_f32: same ids / counts / means / node offsets as sonet_som_assign_sort_f32, every node's run holds the same
    columns, in an order that is the same in every run -- and that IS the stable order inside a 512-point workgroup for k = 1."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + N + M)
    x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(DEV)
    sn = torch.nn.functional.normalize(torch.randn(B, 3, N, generator=g), dim=1).to(DEV)
    node = (torch.rand(B, 3, M, generator=g) * 2 - 1).to(DEV)
    a0, g0 = ops.som_assign_sort(x, sn, node, k)
    a1, g1 = ops.som_assign_sort(x, sn, node, k, deterministic=True)
    a2, g2 = ops.som_assign_sort(x, sn, node, k, deterministic=True)
    assert torch.equal(a0.min_idx_i32, a1.min_idx_i32) and torch.equal(a0.count, a1.count)
    for key in ("ids_sorted", "node_off", "row_max"):
        assert torch.equal(g0[key], g1[key]), key
    assert_close_rms(g1["som_node"].cpu().numpy(), g0["som_node"].cpu().numpy(), 1e-6, "som_node")
    for key in ("x_aug_sorted", "ids_sorted", "pos0", "som_node"):
        assert torch.equal(g1[key], g2[key]), key                 # the same sorted copy in every run
    # the same multiset of columns per node (sorting every node's run by its channel values makes the two copies equal)
    kN = k * N
    for gg in (g0, g1):
        gg["_key"] = gg["ids_sorted"].double() * 1e6 + gg["x_aug_sorted"][:, 3].double() * 1e3 + gg["x_aug_sorted"][:, 4].double()
    o0, o1 = torch.argsort(g0["_key"], dim=1), torch.argsort(g1["_key"], dim=1)
    s0 = torch.gather(g0["x_aug_sorted"], 2, o0.unsqueeze(1).expand(B, 6, kN))
    s1 = torch.gather(g1["x_aug_sorted"], 2, o1.unsqueeze(1).expand(B, 6, kN))
    assert torch.equal(s0, s1)
    if k == 1 and N <= 512:
        # one workgroup, one copy per point: wave / slot / lane order = slot p (points tid, then 256 + tid) ... every node's run lists its
        # points of the first 256 in ascending order, then those of the second 256
        ids = a1.min_idx_i32.long()
        order = torch.argsort(ids * 2 + (torch.arange(N, device=DEV) >= 256).long().unsqueeze(0), dim=1, stable=True)
        ref = torch.gather(sn, 2, order.unsqueeze(1).expand(B, 3, N))
        assert torch.equal(g1["x_aug_sorted"][:, 3:], ref)


def test_sorted_training_forward_is_bit_reproducible():
    """With the deterministic sort the f32-class training step on node-sorted columns gives the same loss, pooled map, gradients and running
    statistics in every run, bit for bit."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 6, 2200
    outs = []
    with ops.precision("h3"):
        for _ in range(2):
            opt = _opt(B, N)
            enc, cls = NW.Encoder(opt), NW.Classifier(opt)
            enc.want_first_pn_out = False
            synth.fill_state_dict_(enc.state_dict(), 3)
            synth.fill_state_dict_(cls.state_dict(), 4)
            enc.to(DEV).train()
            cls.to(DEV).train()
            inp = synth.make_inputs(B, N, seed=9, device=torch.device(DEV))
            with ops.kernel_timing() as rec:
                feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
                loss = torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"])
                loss.backward()
            assert any(n == "som_assign_sort_det" for n, _, _ in rec.records) and any(n.startswith("pointmlph3_segpool") for n, _, _ in rec.records)
            outs.append((loss.detach().clone(), enc.first_pn_out_masked_max.detach().clone(),
                         {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k
    for k in outs[0][3]:
        assert torch.equal(outs[0][3][k], outs[1][3][k]), k
