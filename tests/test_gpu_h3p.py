"""GPU parity tests of the third-generation fp16-split layer (so-net_amd/csrc/pointmlp_h3p.hip): pre-split "P16" activation planes.

Reference semantics: EquivariantLayer.forward / the 1x1 Conv2d layers of KNNModule (models/layers.py:282-296, :313-367) with the
eval-mode BatchNorm and bias folded to (scale, shift).  Checked against the CPU oracle (oracle/cpu_oracle.py::pointwise_layer, which
calls the aten conv the reference calls) and against float64 on the device, float bound 1e-5 * max(|ref|, rms(ref)) as everywhere."""
import numpy as np
import pytest
import torch

from conftest import assert_close_rms

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def cu(t):
    return t.to(DEV).contiguous()


def p16_decode_host(raw, B, C, L):
    """The layout contract of include/sonet_hip.h, restated on the host: raw bytes -> f64 B x C x L."""
    KC = (C + 15) // 16
    a = raw.cpu().numpy().view(np.float16).reshape(B, KC, 2, 2, L, 8).astype(np.float64)
    out = np.zeros((B, KC * 16, L))
    for h in range(2):
        for e in range(8):
            ch = 4 * h + (e & 3) + 8 * (e >> 2)
            out[:, ch::16, :] = (a[:, :, 0, h, :, e] + a[:, :, 1, h, :, e]) / 32.0
    return out[:, :C]


@pytest.mark.parametrize("B,C,L", [(2, 64, 1000), (3, 9, 77), (1, 393, 300), (2, 16, 1), (1, 100, 4097)])
def test_p16_round_trip_and_layout(B, C, L):
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C + L)
    x = torch.randn(B, C, L, generator=g) * torch.logspace(-2, 2, C).view(1, C, 1)
    p = ops.p16_from_f32(cu(x))
    assert p.data.numel() == B * ((C + 15) // 16) * 64 * L
    host = p16_decode_host(p.data, B, C, L)
    # two fp16 pieces: 22 significand bits, and 32 x the residual stays a normal fp16 down to |x| ~ 4e-3 x the value's hi piece
    err = np.abs(host - x.double().numpy())
    assert (err <= 2.0 ** -21 * np.abs(x.double().numpy()) + 2.0 ** -30).all()
    # the device decode is the same arithmetic
    np.testing.assert_array_equal(ops.p16_to_f32(p).cpu().numpy(), host.astype(np.float32))
    # padded channels are zero planes
    KC = (C + 15) // 16
    raw = p.data.cpu().numpy().view(np.float16).reshape(B, KC, 2, 2, L, 8)
    for h in range(2):
        for e in range(8):
            ch = (KC - 1) * 16 + 4 * h + (e & 3) + 8 * (e >> 2)
            if ch >= C:
                assert not raw[:, KC - 1, :, h, :, e].any()


def test_p16_affine_relu_clamp_and_range_log():
    from sonet_hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 40, 333, generator=g)
    sc, sh = torch.rand(40, generator=g) + 0.5, torch.randn(40, generator=g)
    p = ops.p16_from_f32(cu(x), cu(sc), cu(sh), relu=True)
    ref = torch.relu(x.double() * sc.double().view(1, -1, 1) + sh.double().view(1, -1, 1))
    got = ops.p16_to_f32(p).double().cpu()
    assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    # values beyond the fp16-split range are clamped to +-2047 and the launch says so in its range-log slot
    big = x.clone()
    big[0, 3, 7] = 5000.0
    big[1, 5, 9] = -1e6
    with ops.range_scope(torch.device(DEV)) as rs:
        pb = ops.p16_from_f32(cu(big))
    bad = rs.violations()
    assert bad and "2047" in bad[0][1]
    dec = ops.p16_to_f32(pb).cpu()
    assert float(dec[0, 3, 7]) == 2047.0 and float(dec[1, 5, 9]) == -2047.0
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.p16_from_f32(cu(x))
    assert rs.violations() == []


def _layer_case(B, C1, C2, Cout, L, relu, seed, scale_x=1.0):
    g = torch.Generator().manual_seed(seed)
    Cin = C1 + C2
    x = torch.randn(B, Cin, L, generator=g) * scale_x
    W = torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5
    bias = torch.rand(Cout, generator=g) * 0.2 - 0.1
    gamma, beta = 0.5 + torch.rand(Cout, generator=g), torch.rand(Cout, generator=g) * 0.4 - 0.2
    mean, var = 0.2 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g)
    return x, W, bias, (gamma, beta, mean, var)


# the dispatch picks the tile shape from the launch size: the small cases run 2x1 / 4x1 tiles, the large ones 4x2, 6x2 and 8x2
@pytest.mark.parametrize("B,C1,C2,Cout,L,relu", [
    (2, 6, 0, 64, 300, True), (2, 64, 0, 128, 768, True), (1, 64, 256, 384, 1000, False), (3, 387, 0, 512, 576, True),
    (4, 515, 0, 768, 64, True), (2, 768, 0, 1024, 64, False), (1, 3, 0, 32, 1, True), (2, 384, 9, 1024, 333, True),
    (2, 128, 0, 96, 77, False), (3, 64, 0, 256, 50000, True), (2, 64, 256, 384, 70000, False), (2, 128, 0, 128, 140001, True),
    (5, 1024, 0, 512, 3072, True)])
def test_pointmlp_h3p_vs_oracle(B, C1, C2, Cout, L, relu):
    from oracle import cpu_oracle as O
    from sonet_hip import ops
    x, W, bias, (gamma, beta, mean, var) = _layer_case(B, C1, C2, Cout, L, relu, C1 + Cout + L)
    scale = gamma / torch.sqrt(var + 1e-5)
    shift = (bias - mean) * scale + beta
    small = B * L <= 20000
    if small:
        ref = O.pointwise_layer(x.numpy(), W.numpy(), bias.numpy(), bn=(gamma.numpy(), beta.numpy(), mean.numpy(), var.numpy()), relu=relu)
    else:                                  # sizes the CPU oracle does not finish in seconds: float64 on the device (same formula)
        r = torch.einsum("oc,bcl->bol", cu(W).double(), cu(x).double()) * cu(scale).double().view(1, -1, 1) + cu(shift).double().view(1, -1, 1)
        ref = (torch.relu(r) if relu else r).cpu().numpy()
    wp = ops.pointmlp_h3p_pack(cu(W))
    x1 = ops.p16_from_f32(cu(x[:, :C1]))
    x2 = ops.p16_from_f32(cu(x[:, C1:])) if C2 else None
    y, yp = ops.pointmlp_h3p(x1, wp, cu(scale), cu(shift), relu, Cout, x2=x2, out="both")
    assert_close_rms(y.cpu().numpy(), ref, 1e-5, "pointmlp_h3p f32 output")
    # the P16 output is the split of the same values (22 bits)
    assert_close_rms(ops.p16_to_f32(yp).cpu().numpy(), y.cpu().numpy(), 1e-6, "pointmlp_h3p P16 output")
    # single-output launches are other kernel instantiations: bit-identical to the "both" launch
    assert torch.equal(ops.pointmlp_h3p(x1, wp, cu(scale), cu(shift), relu, Cout, x2=x2, out="f32"), y)
    assert torch.equal(ops.pointmlp_h3p(x1, wp, cu(scale), cu(shift), relu, Cout, x2=x2, out="p16").data, yp.data)
    # and close to the second generation on the same operands (same products, other summation order inside the MFMA)
    if Cout % 32 == 0 and (C2 == 0 or C1 % 16 == 0):
        wp2 = ops.pointmlp_pack(cu(W), "h3")
        y2 = ops.pointmlp(cu(x[:, :C1]), wp2, cu(scale), cu(shift), relu, Cout, x2=cu(x[:, C1:]) if C2 else None)
        assert_close_rms(y.cpu().numpy(), y2.cpu().numpy(), 1e-5, "third vs second generation")


def test_pointmlp_h3p_chain_stays_f32_class():
    """Three layers handed on in P16 (the segmenter tail's shapes, 1024 -> 512 -> 256 -> 128): the end result meets float64 at 1e-5."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(77)
    B, L = 2, 3000
    chans = [1024, 512, 256, 128]
    x = torch.rand(B, chans[0], L, generator=g)
    cur = ops.p16_from_f32(cu(x))
    ref = cu(x).double()
    for i in range(3):
        W = torch.randn(chans[i + 1], chans[i], generator=g) * (2.0 / chans[i]) ** 0.5
        sc, sh = torch.rand(chans[i + 1], generator=g) + 0.5, torch.randn(chans[i + 1], generator=g) * 0.1
        last = i == 2
        cur = ops.pointmlp_h3p(cur, ops.pointmlp_h3p_pack(cu(W)), cu(sc), cu(sh), True, chans[i + 1], out="f32" if last else "p16")
        ref = torch.relu(torch.einsum("oc,bcl->bol", cu(W).double(), ref) * cu(sc).double().view(1, -1, 1) + cu(sh).double().view(1, -1, 1))
    assert_close_rms(cur.cpu().numpy(), ref.cpu().numpy(), 1e-5, "three layers through P16")


@pytest.mark.parametrize("M,K,C", [(64, 9, 384), (64, 9, 64), (30, 5, 128)])
def test_pointmlp_h3p_gather(M, K, C):
    """x1 read through a per-column gather index (KNNModule's neighbour gather done by the operand loads, models/layers.py:313-350)
    == the layer on the gathered tensor; out-of-range indices read zeros."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(M + K + C)
    B, Cout, L = 3, 512, M * K
    feat = torch.randn(B, C, M, generator=g)
    small = torch.randn(B, 3, L, generator=g)
    gidx = torch.randint(-1, M + 1, (B, L), generator=g, dtype=torch.int32)
    W = torch.randn(Cout, C + 3, generator=g) * (2.0 / (C + 3)) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    ok = ((gidx >= 0) & (gidx < M)).unsqueeze(1)
    gathered = torch.gather(feat, 2, gidx.clamp(0, M - 1).long().unsqueeze(1).expand(B, C, L)) * ok
    ref = torch.relu(torch.einsum("oc,bcl->bol", W.double(), torch.cat([gathered, small], 1).double()) * sc.double().view(1, -1, 1)
                     + sh.double().view(1, -1, 1))
    wp = ops.pointmlp_h3p_pack(cu(W))
    y = ops.pointmlp_h3p(ops.p16_from_f32(cu(feat)), wp, cu(sc), cu(sh), True, Cout, x2=ops.p16_from_f32(cu(small)), gidx=cu(gidx))
    assert_close_rms(y.cpu().numpy(), ref.numpy(), 1e-5, "gathered layer")
    direct = ops.pointmlp_h3p(ops.p16_from_f32(cu(gathered)), wp, cu(sc), cu(sh), True, Cout, x2=ops.p16_from_f32(cu(small)))
    assert torch.equal(y, direct)


@pytest.mark.parametrize("B,C1,C2,Cout,L,M", [(3, 384, 9, 1024, 700, 64), (2, 64, 0, 256, 33, 5), (4, 128, 6, 512, 3072, 64), (64, 384, 9, 1024, 3072, 64)])
def test_pointmlp_h3p_nodeadd(B, C1, C2, Cout, L, M):
    """Per-node addend in the epilogue (segmenter layer 1, models/networks.py:296-326): == float64 of (W x + z[node]) * scale + shift."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = cu(torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5)
    x1 = cu(torch.randn(B, C1, L, generator=g))
    x2 = cu(torch.randn(B, C2, L, generator=g)) if C2 else None
    z = cu(torch.randn(B, Cout, M, generator=g))
    idx = cu(torch.randint(-1, M + 1, (B, L), generator=g, dtype=torch.int32))
    scale, shift = cu(torch.rand(Cout, generator=g) + 0.5), cu(torch.randn(Cout, generator=g))
    wp = ops.pointmlp_h3p_pack(w)
    p1, p2 = ops.p16_from_f32(x1), (ops.p16_from_f32(x2) if C2 else None)
    got = ops.pointmlp_h3p(p1, wp, scale, shift, True, Cout, x2=p2, z=z, zidx=idx)
    ok = (idx >= 0) & (idx < M)
    nb = min(B, 4)                                       # (float64 of the largest case does not fit comfortably: check the first clouds)
    xin = (torch.cat([x1, x2], 1) if C2 else x1)[:nb].double()
    zg = torch.gather(z[:nb].double(), 2, idx[:nb].clamp(0, M - 1).long().unsqueeze(1).expand(nb, Cout, L)) * ok[:nb].unsqueeze(1)
    ref = torch.relu((torch.einsum("oc,bcl->bol", w.double(), xin) + zg) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1))
    assert_close_rms(got[:nb].cpu().numpy(), ref.cpu().numpy(), 1e-5, "node-add layer")
    gp = ops.pointmlp_h3p(p1, wp, scale, shift, True, Cout, x2=p2, z=z, zidx=idx, out="p16")
    assert_close_rms(ops.p16_to_f32(gp).cpu().numpy(), got.cpu().numpy(), 1e-6, "node-add layer, P16 output")


@pytest.mark.parametrize("B,C1,C2,Cout,L", [(3, 64, 0, 128, 1500), (2, 6, 0, 64, 777), (2, 256, 64, 384, 300), (5, 128, 0, 256, 33), (8, 64, 256, 384, 15000)])
def test_pointmlp_h3p_statistics_epilogue(B, C1, C2, Cout, L):
    """Training forward: the same output as the plain launch bit for bit, mean / biased variance to 1e-6 of the channel's scale."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = cu(torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5)
    x1 = ops.p16_from_f32(cu(torch.randn(B, C1, L, generator=g)))
    x2 = ops.p16_from_f32(cu(torch.randn(B, C2, L, generator=g))) if C2 else None
    scale = cu(torch.ones(Cout))
    shift = cu(torch.randn(Cout, generator=g) * 3.0)
    wp = ops.pointmlp_h3p_pack(w)
    y0 = ops.pointmlp_h3p(x1, wp, scale, shift, False, Cout, x2=x2)
    y1, m1, v1 = ops.pointmlp_h3p(x1, wp, scale, shift, False, Cout, x2=x2, stats=True)
    assert torch.equal(y0, y1)
    ref = y0.double()
    mref, vref = ref.mean(dim=(0, 2)), ref.var(dim=(0, 2), unbiased=False)
    sc = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
    assert float(((m1.double() - mref).abs() / sc).max()) < 1e-6
    assert float(((v1.double() - vref).abs() / sc ** 2).max()) < 2e-6


def test_pointmlp_h3p_output_range_is_logged():
    """A P16-producing launch reports its largest post-activation magnitude: its consumers cannot check the clamp any more."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(3)
    x = ops.p16_from_f32(cu(torch.randn(2, 64, 500, generator=g)))
    W = cu(torch.randn(128, 64, generator=g))
    one, zero = cu(torch.ones(128)), cu(torch.zeros(128))
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.pointmlp_h3p(x, ops.pointmlp_h3p_pack(W), one, zero, True, 128, out="p16")
    assert rs.violations() == []
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.pointmlp_h3p(x, ops.pointmlp_h3p_pack(W), one * 1000.0, zero, True, 128, out="p16")
    bad = rs.violations()
    assert bad and "2047" in bad[0][1]
    with ops.range_scope(torch.device(DEV)) as rs:           # an f32 output has no range (nothing was clamped)
        ops.pointmlp_h3p(x, ops.pointmlp_h3p_pack(W), one * 1000.0, zero, True, 128, out="f32")
    assert rs.violations() == []
    with ops.range_scope(torch.device(DEV)) as rs:           # weights beyond the fp16(32 w) range
        ops.pointmlp_h3p(x, ops.pointmlp_h3p_pack(W * 4000.0), one, zero, True, 128, out="f32")
    assert rs.violations()


@pytest.mark.gpu
@pytest.mark.parametrize("Cout,L,row", [(512, 500, 500), (1024, 3072, 1000), (512, 3072, 200)])
def test_pointmlp_h3p_output_range_is_logged_from_every_output_slab(Cout, L, row):
    """The production layers split Cout over several output slabs (one workgroup set per slab): an out-of-range value in a HIGH
    channel -- a slab other than slab 0 -- must reach the range log as well (only slab 0 used to publish)."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(5)
    x = ops.p16_from_f32(cu(torch.randn(2, 64, L, generator=g)))
    W = cu(torch.randn(Cout, 64, generator=g))
    one, zero = cu(torch.ones(Cout)), cu(torch.zeros(Cout))
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.pointmlp_h3p(x, ops.pointmlp_h3p_pack(W), one, zero, True, Cout, out="p16")
    assert rs.violations() == []
    sc = one.clone()
    sc[row] = 1000.0                                         # only this channel leaves +-2047
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.pointmlp_h3p(x, ops.pointmlp_h3p_pack(W), sc, zero, True, Cout, out="p16")
    bad = rs.violations()
    assert bad and "2047" in bad[0][1]


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,L,M", [(2, 384, 3072, 64), (3, 40, 777, 7), (1, 16, 64, 64), (64, 384, 3072, 64)])
def test_index_max_gather_on_p16_planes_equals_the_f32_kernel_on_the_decoded_values(B, C, L, M):
    """sonet_index_max_gather_p16 (the per-node arg-max pool straight on P16 planes) == sonet_index_max_gather_f32 on p16_to_f32 of the same
    planes, bit for bit (positions and values), incl. empty nodes (position 0), masked nodes, out-of-range ids, ties, +-0, a padded last chunk."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + C + L)
    x = torch.randn(B, C, L, generator=g)
    x[:, :, 5] = x[:, :, 3]                                     # ties: the first wins
    x[0, 0, :] = 0.0
    x[0, 1, ::2] = -0.0
    idx = torch.randint(0, M, (B, L), generator=g, dtype=torch.int32)
    if M > 3:
        idx[idx == 2] = 3                                       # node 2 is empty
        idx[0, :10] = M + 5                                     # ids nobody owns
    row_max = torch.ones(B, M, dtype=torch.int32)
    row_max[:, 1 % M] = 0                                       # a masked node gathers position 0
    p = ops.p16_from_f32(x.to(DEV))
    dec = ops.p16_to_f32(p)
    i0, v0 = ops.index_max_gather(dec, idx.to(DEV), M, row_max.to(DEV))
    i1, v1 = ops.index_max_gather_p16(p, idx.to(DEV), M, row_max.to(DEV))
    assert torch.equal(i0, i1)
    assert torch.equal(v0.view(torch.int32), v1.view(torch.int32))


@pytest.mark.gpu
def test_fused_first_pointnet_p16_only_equals_the_variant_that_also_stores_f32():
    """sonet_pointresnet_fused_p16_f32 with y == NULL writes the same planes as with y, and the segmenter's forward through them (P16-only
    first PointNet + index_max_gather_p16) stays within 1e-5 of the path that stores first_pn_out in f32."""
    from argparse import Namespace
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 4, 1024
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024, activation="relu",
                    normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="center", bn_momentum=0.1,
                    bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=50)
    enc, seg = NW.Encoder(opt), NW.Segmenter(opt)
    synth.fill_state_dict_(enc.state_dict(), 1)
    synth.fill_state_dict_(seg.state_dict(), 2)
    enc.to(DEV).eval()
    seg.to(DEV).eval()
    inp = synth.make_inputs(B, N, seed=3, device=torch.device(DEV))
    label = torch.randint(0, 16, (B,), device=DEV)
    outs, names = {}, {}
    for only in (True, False):
        ops.P16_ONLY = only
        try:
            with torch.no_grad(), ops.kernel_timing() as rec:
                outs[only] = NW.segmentation_forward(enc, seg, inp["pc"], inp["sn"], label, inp["node"], inp["node_knn_I"]).clone()
                planes = enc.first_pointnet.last_p16.data.clone()
                first = enc.first_pn_out.clone()                # (decoded from the planes when only they were written)
                pooled = enc.first_pn_out_masked_max.clone()
        finally:
            ops.P16_ONLY = True
        names[only] = set(n for n, _, _ in rec.records)
        outs[(only, "planes")], outs[(only, "first")], outs[(only, "pooled")] = planes, first, pooled
    assert any(n.startswith("pointresnet_fused_p16only") for n in names[True]) and "index_max_gather_p16" in names[True]
    assert any(n.startswith("pointresnet_fused_p16_") for n in names[False]) and "index_max_gather" in names[False]
    assert torch.equal(outs[(True, "planes")], outs[(False, "planes")])
    scale = float(outs[(False, "first")].abs().max())
    assert float((outs[(True, "first")] - outs[(False, "first")]).abs().max()) <= 2.0 ** -21 * scale        # 22-bit planes vs the f32 store
    assert float((outs[(True, "pooled")] - outs[(False, "pooled")]).abs().max()) <= 2.0 ** -21 * scale
    ref = outs[False]
    assert float((outs[True] - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0)
