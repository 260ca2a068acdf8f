"""GPU tests of the VARIANTS build (libsonet_hip_variants.so: the SONET_* tuning knobs and the measured-slower kernels kept as records).
Not collected by the main run (the file name does not match test_*): tests/test_gpu_variants_suite.py runs it in a subprocess with
SONET_HIP_LIB pointing at the variants library, so that the product process never loads it.

    SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so python -m pytest tests/variants/variants_gpu.py -q
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import ROOT, assert_close_rms, golden  # noqa: E402,F401

from sonet_hip import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

if os.path.abspath(_lib.LIB_PATH) != os.path.abspath(_lib.VARIANTS_PATH):
    pytest.skip("needs SONET_HIP_LIB=%s" % _lib.VARIANTS_PATH, allow_module_level=True)

from sonet_hip import variants  # noqa: E402


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


@pytest.mark.parametrize("mt", ["12", "6", "4", "2", "1"])
def test_pointmlp_bf16_tile_variants_agree(mt, monkeypatch):
    from sonet_hip import ops
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(2, 64, 700, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(2, 256, 700, generator=g).to(torch.bfloat16).to(DEV)
    W = (torch.randn(384, 320, generator=g) * 0.08).to(DEV)
    wp = ops.pointmlp_pack(W, "bf16")
    one, zero = ops.const_vec(384, 1.0, DEV), ops.const_vec(384, 0.0, DEV)
    base = ops.pointmlp(x1, wp, one, zero, False, 384, x2=x2).clone()
    monkeypatch.setenv("SONET_BF16_MT", mt)
    for s in ("1", "2"):
        monkeypatch.setenv("SONET_BF16_S", s)
        assert torch.equal(ops.pointmlp(x1, wp, one, zero, False, 384, x2=x2), base), (mt, s)   # same K order: bit-identical




@pytest.mark.parametrize("B,C1,C2,Cout,L", [(3, 6, 0, 64, 1500), (2, 64, 0, 128, 3000), (5, 128, 0, 256, 1500), (3, 64, 256, 384, 1500),
                                            (2, 64, 256, 384, 130), (2, 256, 0, 128, 700), (2, 384, 0, 64, 900), (1, 320, 0, 384, 2), (70, 64, 256, 384, 64)])
def test_pointmlp_bf16_xreg_equals_streaming(B, C1, C2, Cout, L, monkeypatch):
    """The X-in-registers kernel (big launches) and the streaming kernel add the K chunks in the same order: bit-identical."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(L + C1)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    W = (torch.randn(Cout, C1 + C2, generator=g) * 0.1).to(DEV)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    wp = ops.pointmlp_pack(W, "bf16")
    monkeypatch.setenv("SONET_BF16_XREG", "0")
    base = ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2).clone()
    monkeypatch.setenv("SONET_BF16_XREG", "2")
    with ops.kernel_timing():
        got = ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2)
    assert torch.equal(got, base)




@pytest.mark.parametrize("B,Na,Nb", [(8, 1280, 5000), (3, 5000, 1280), (2, 1, 1), (2, 300, 300), (1, 2049, 4100), (2, 4097, 17)])
def test_chamfer_nn2_equals_two_one_direction_searches(B, Na, Nb):
    """Row and column arg-min from ONE sweep of the distance matrix == two sonet_chamfer_nn_f32 launches == the oracle's exact
    search (bit-exact indices, ties -> lowest index; duplicated points force ties)."""
    from oracle import cpu_oracle as O
    from sonet_hip import ops
    g = torch.Generator().manual_seed(Na + Nb)
    a = torch.rand(B, 3, Na, generator=g) * 2 - 1
    b = torch.rand(B, 3, Nb, generator=g) * 2 - 1
    if Na > 10 and Nb > 10:
        a[:, :, 5] = a[:, :, 3]                     # exact ties in both directions
        b[:, :, 7] = b[:, :, 2]
    ab, ba = variants.chamfer_nn2(a.to(DEV), b.to(DEV))
    np.testing.assert_array_equal(ab.cpu().numpy(), ops.chamfer_nn(a.to(DEV), b.to(DEV)).cpu().numpy())
    np.testing.assert_array_equal(ba.cpu().numpy(), ops.chamfer_nn(b.to(DEV), a.to(DEV)).cpu().numpy())
    np.testing.assert_array_equal(ab.cpu().numpy(), O.chamfer_nn(a.numpy(), b.numpy()))
    np.testing.assert_array_equal(ba.cpu().numpy(), O.chamfer_nn(b.numpy(), a.numpy()))




@pytest.mark.parametrize("B,C1,C2,Cout,L,L1", [(4, 515, 0, 768, 64, 0), (3, 387, 0, 512, 576, 0), (2, 384, 3, 512, 576, 64),
                                              (2, 256, 64, 384, 700, 0), (3, 40, 0, 128, 77, 0), (2, 17, 0, 128, 1, 0),
                                              (1, 1347, 0, 512, 1029, 0), (5, 16, 0, 256, 33, 0)])
def test_pointmlp_h3_second_generation_is_bit_identical(B, C1, C2, Cout, L, L1, monkeypatch):
    """pointmlp_h3r_kernel (LDS-DMA ring, split in the MFMA shadow, XCD-aware slab order) == the first-generation fp16-split
    kernel bit for bit -- same MFMAs in the same order per accumulator -- including K tails (Cin % 32 != 0), ragged column
    tiles, the two-input and the gather forms; and both meet float64 within the f32-class bound."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = (torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(DEV)
    x1 = torch.randn(B, C1, L1 if L1 else L, generator=g).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(DEV) if C2 else None
    gidx = torch.randint(-1, L1 + 1, (B, L), generator=g, dtype=torch.int32).to(DEV) if L1 else None
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    wp = ops.pointmlp_pack(w, "h3")
    out = {}
    for gen in ("1", "0"):
        monkeypatch.setenv("SONET_POINTMLP_H3R", gen)
        with ops.kernel_timing():
            out[gen] = ops.pointmlp(x1, wp, scale, shift, True, Cout, x2=x2, gidx=gidx)
    assert torch.equal(out["1"], out["0"])
    xa = x1.double()
    if gidx is not None:
        ok = (gidx >= 0) & (gidx < L1)
        xa = torch.gather(xa, 2, gidx.clamp(0, L1 - 1).long().unsqueeze(1).expand(B, C1, L)) * ok.unsqueeze(1)
    xin = torch.cat([xa, x2.double()], 1) if x2 is not None else xa
    ref = torch.relu(torch.einsum("oc,bcl->bol", w.double(), xin) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1))
    err = (out["1"].double() - ref).abs()
    assert float(err.max()) <= 1e-5 * max(1.0, float(ref.abs().max()))




@pytest.mark.parametrize("B,C,Cout,M,K", [(3, 512, 512, 64, 9), (2, 64, 128, 5, 3), (1, 128, 256, 100, 16)])
def test_pointmlp_kmax_epilogue(B, C, Cout, M, K):
    """Layer + max over the K planes of its k-major output from the epilogue == planes_max(pointmlp(...)) bit for bit (max is exact,
    the layer kernel is the same), incl. negative outputs (no ReLU)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C + Cout + M)
    w = cu(torch.randn(Cout, C, generator=g) / C ** 0.5)
    x = cu(torch.randn(B, C, K * M, generator=g))
    scale, shift = cu(torch.rand(Cout, generator=g) + 0.5), cu(torch.randn(Cout, generator=g))
    wp = ops.pointmlp_pack(w, "h3")
    for relu in (True, False):
        ref = ops.planes_max(ops.pointmlp(x, wp, scale, shift, relu, Cout), K)
        got = variants.pointmlp_kmax(x, wp, scale, shift, relu, Cout, M)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("B,C,M,C1,C2,L", [(3, 384, 64, 64, 256, 3000), (2, 96, 8, 16, 48, 130), (2, 48, 7, 44, 0, 257), (8, 384, 64, 64, 256, 15000)])
def test_sparse_input_gradient_kernels_agree(B, C, M, C1, C2, L, monkeypatch):
    """pooled_dgrad: the one-channel kernel (SONET_PD_ONE=1), its four-channel column-owned twin (SONET_PD_KERNEL=4: the same bits) and the
    product's entry-balanced kernel (5: the same sums, a cut column associates differently -- f32 rounding of a handful of terms), incl. positions
    piled on few columns (chunks that are one column from end to end) and a partial channel slab."""
    from sonet_hip import ops
    gen = torch.Generator().manual_seed(B + L)
    g = torch.randn(B, C, M, generator=gen)
    pos = torch.randint(0, L, (B, C, M), generator=gen, dtype=torch.int32)
    pos[:, : C // 2, : max(M // 2, 1)] = 5                                 # hundreds of entries on one column
    pos[:, C // 2:, 0] = L - 1
    W = torch.randn(C, C1 + C2, generator=gen) * 0.1
    outs = {}
    for name, env in (("one", {"SONET_PD_ONE": "1"}), ("k4", {"SONET_PD_KERNEL": "4"}), ("k5", {"SONET_PD_KERNEL": "5"}), ("default", {})):
        for k in ("SONET_PD_ONE", "SONET_PD_KERNEL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        a = ops.pooled_dgrad(g.to(DEV), pos.to(DEV), W.to(DEV), C1, C2, L)
        b = ops.pooled_dgrad(g.to(DEV), pos.to(DEV), W.to(DEV), C1, C2, L)
        outs[name] = torch.cat([t for t in a if t is not None and t.numel()], dim=1)
        assert torch.equal(outs[name], torch.cat([t for t in b if t is not None and t.numel()], dim=1))       # run to run
    assert torch.equal(outs["one"], outs["k4"])
    assert torch.equal(outs["default"], outs["k5"])
    G = torch.zeros(B, C, L, dtype=torch.float64).scatter_add_(2, pos.long(), g.double())
    ref = torch.matmul(W.double().t().unsqueeze(0), G)
    for name in ("k4", "k5"):
        assert float((outs[name].cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), name


# ------------------------------------------------------------------------------------------ measured-slower records of round 5
# (moved here from tests/test_gpu_segpool.py in round 6: the kernels live in the variants build only, as DESIGN section 1 requires)
from argparse import Namespace  # noqa: E402


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



@pytest.mark.parametrize("B,C,M,C1,C2,L", [(64, 384, 64, 64, 256, 15000), (3, 384, 64, 64, 256, 3000), (2, 96, 8, 16, 48, 130), (2, 48, 7, 44, 0, 257)])
def test_pooled_dgrad_with_its_tail_equals_dgrad_then_scatter_and_statistics(B, C, M, C1, C2, L):
    """sonet_pooled_dgrad_tail_f32: the gradients == sonet_pooled_dgrad_f32 followed by the addition of the column-0 part at pos0 (bit for
    bit: the same f32 add), the BatchNorm-backward sums of the layer below == float64 sums over the same gradient and mask (and the
    statistics pass on it); run to run identical; incl. a ragged last tile, a channel slab that mixes the two panels and no second panel."""
    from sonet_hip import ops
    gen = torch.Generator().manual_seed(B + L + C2)
    g = (torch.randn(B, C, M, generator=gen) * 1e-2).to(DEV)
    pos = torch.randint(0, L, (B, C, M), generator=gen, dtype=torch.int32)
    pos[:, : C // 4, 0] = L - 1                                          # a pile on the last column of the ragged tile
    pos[:, 0, 1] = -1
    pos = pos.to(DEV)
    W = (torch.randn(C, C1 + C2, generator=gen) * 0.1).to(DEV)
    col0 = (torch.randn(B, C1 + C2, generator=gen) * 1e-2).to(DEV)
    p0 = torch.randint(0, L, (B,), generator=gen, dtype=torch.int32)
    p0[0] = L - 1
    p0 = p0.to(DEV)
    r1, r2 = ops.pooled_dgrad(g, pos, W, C1, C2, L)
    p0l = p0.long().view(B, 1, 1)
    r1.scatter_add_(2, p0l.expand(B, C1, 1), col0[:, :C1].unsqueeze(2))
    if C2:
        r2.scatter_add_(2, p0l.expand(B, C2, 1), col0[:, C1:].unsqueeze(2))
    a1, a2 = ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0, pos0=p0)
    assert torch.equal(a1, r1) and (C2 == 0 or torch.equal(a2, r2))
    if C2 == 0:
        return
    raw = (torch.randn(B, C2, L, generator=gen) * 1.5).to(DEV)
    sc, sh = (torch.rand(C2, generator=gen) + 0.5).to(DEV), (torch.randn(C2, generator=gen) * 0.3).to(DEV)
    for relu in (True, False):
        b1, b2, sums = ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0, pos0=p0, below=(raw, sc, sh, relu))
        assert torch.equal(b1, r1) and torch.equal(b2, r2)
        keep = (raw.double() * sc.double().view(1, C2, 1) + sh.double().view(1, C2, 1)) > 0 if relu else torch.ones_like(raw, dtype=torch.bool)
        gm = torch.where(keep, r2.double(), torch.zeros((), dtype=torch.float64, device=DEV))
        ref = torch.cat((gm.sum(dim=(0, 2)), (gm * raw.double()).sum(dim=(0, 2))))
        scale = torch.cat((gm.abs().sum(dim=(0, 2)), (gm * raw.double()).abs().sum(dim=(0, 2)))) + 1e-30
        assert float(((sums - ref).abs() / scale).max()) <= 2e-6
        other = ops.pointwise_bwd_stats(r2, raw, sc, sh, relu, want_sums=True)
        assert float(((sums - other).abs() / scale).max()) <= 2e-6
        again = ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0, pos0=p0, below=(raw, sc, sh, relu))[2]
        assert torch.equal(again, sums)
    with pytest.raises(ops.SonetHipError):
        ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0)



def test_training_step_with_the_tail_of_the_sparse_input_gradient():
    """ops.POOLED_DGRAD_TAIL on / off on the node-sorted f32-class path: the statistics pass over (gy, raw) of the widest hidden layer and
    the two scatter_add launches are gone, the step is the same to f32-class accuracy (the sums run in another order)."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 8, 3000
    res = {}
    old = ops.POOLED_DGRAD_TAIL
    try:
        with ops.precision("h3"):
            for flag in (True, False):
                ops.POOLED_DGRAD_TAIL = flag
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
                assert ("pooled_dgrad_tail_sums" in names) == flag and ("pooled_dgrad" in names) != flag
                res[flag] = (loss.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                             {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone(),
                             sum(n == "pointwise_bwd_stats" for n in names))
    finally:
        ops.POOLED_DGRAD_TAIL = old
    assert res[True][4] == res[False][4] - 1
    _compare_steps(res[True][:4], res[False][:4])



@pytest.mark.parametrize("B,C,Cout,L", [(64, 256, 128, 15000), (8, 128, 64, 4100), (3, 64, 96, 577)])
def test_dgrad_epilogue_returns_the_batchnorm_backward_sums_of_the_layer_below(B, C, Cout, L):
    """ops.pointmlp_x3_bnb(..., below=(praw, psc, psh, relu)): the launch's output is gy of the layer below; its epilogue returns that layer's
    sums (sum gy * mask, sum gy * mask * praw) as sonet_pointwise_bwd_stats_f32 computes them from one more pass (double accumulation of
    32-column float partials: equal to summation order); output and g_raw are untouched."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C + L)
    gy = (torch.randn(B, C, L, generator=g) * 1e-3).to(DEV)
    raw = (torch.randn(B, C, L, generator=g) * 1.5).to(DEV)
    praw = (torch.randn(B, Cout, L, generator=g) * 1.5 + 0.2).to(DEV)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV)
    psc, psh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    a, b, c0 = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV), (torch.randn(C, generator=g) * 1e-5).to(DEV)
    wpt = ops.pointmlp_pack((torch.randn(Cout, C, generator=g) * C ** -0.5).to(DEV), "x3")
    one, zero = ops.const_vec(Cout, 1.0, DEV), ops.const_vec(Cout, 0.0, DEV)
    y_ref, g_ref = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, True, Cout)
    for prelu in (True, False):
        y, g_raw, sums = ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, True, Cout, below=(praw, psc, psh, prelu))
        assert torch.equal(y, y_ref) and torch.equal(g_raw, g_ref)
        ref = ops.pointwise_bwd_stats(y_ref, praw, psc, psh, prelu, want_sums=True)
        scale = (y_ref.double().abs() * (praw.double().abs() + 1)).sum(dim=(0, 2)).repeat(2)     # the size of the summed terms per channel
        assert float(((sums - ref).abs() / scale).max()) <= 1e-6, prelu
        assert torch.equal(sums, ops.pointmlp_x3_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, True, Cout, below=(praw, psc, psh, prelu))[2])   # fixed order



def test_training_step_with_backward_sums_from_the_dgrad_epilogue():
    """ops.BWD_STATS_EPILOGUE on / off: the statistics launches of the hidden layers whose gradient comes from one dgrad launch disappear;
    forward identical, gradients within the arg-max-flip bound (the sums differ by summation order)."""
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 8, 3000
    res = {}
    old = ops.BWD_STATS_EPILOGUE
    try:
        with ops.precision("h3"):
            for flag in (True, False):
                ops.BWD_STATS_EPILOGUE = flag
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
                assert any(n.startswith("pointmlpx3_bnbs_") for n in names) == flag
                res[flag] = (loss.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                             {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}, enc.first_pn_out_masked_max.detach().clone(),
                             sum(1 for n in names if n == "pointwise_bwd_stats"))
    finally:
        ops.BWD_STATS_EPILOGUE = old
    assert res[True][4] < res[False][4]
    _compare_steps(res[True], res[False])


