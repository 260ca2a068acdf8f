"""No-grad node-level stage on the flat column axis (so-net_amd/csrc/node_stage.hip + the group-max epilogues of pointmlp_h3p.hip):
KNNModule + final PointNet + global max (models/layers.py:313-367,384-387, models/networks.py:187-197) in five launches on pre-split
activations.  Pieces against the kernels they replace (bit for bit where the arithmetic is the same), the whole stage against the
round-4 stage and the CPU oracle (1e-5 * max(|ref|, rms))."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import assert_close_rms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cu(t):
    return t.to(DEV).contiguous()


def _layer(C1, Cout, seed, C2=0):
    g = torch.Generator().manual_seed(seed)
    W = cu(torch.randn(Cout, C1 + C2, generator=g) * (1.0 / np.sqrt(C1 + C2)))
    sc = cu(torch.rand(Cout, generator=g) + 0.5)
    sh = cu(torch.randn(Cout, generator=g) * 0.3)
    return W, sc, sh


@pytest.mark.parametrize("C1,Cout,nblk,K,G,relu", [(512, 512, 5, 9, 14, True), (64, 128, 3, 9, 14, False), (48, 96, 2, 16, 8, True),
                                                    (32, 64, 4, 1, 16, True), (64, 64, 1, 5, 16, True), (16, 32, 2, 128, 1, True)])
def test_gmax_p16_equals_layer_then_group_max(C1, Cout, nblk, K, G, relu):
    """sonet_pointmlp_h3p_gmax (P16 out) == the same layer with its columns written (sonet_pointmlp_h3p, f32) followed by the max over every
    group's K columns and the split: bit for bit (same accumulators, same affine, the max of identical values)."""
    from sonet_hip import ops
    L = nblk * 128
    g = torch.Generator().manual_seed(C1 + Cout + K)
    x = torch.randn(1, C1, L, generator=g)
    xp = ops.p16_from_f32(cu(x))
    W, sc, sh = _layer(C1, Cout, 3)
    wp = ops.pointmlp_h3p_pack(W)
    ngout = nblk * G - 3 if nblk * G > 3 else nblk * G            # (the last groups of the last block are not outputs)
    Lout = (ngout + 127) // 128 * 128
    got = ops.pointmlp_h3p_gmax(xp, wp, sc, sh, relu, Cout, K, G, ngout, out="p16", Lout=Lout)
    y = ops.pointmlp_h3p(xp, wp, sc, sh, relu, Cout, out="f32")[0]               # Cout x L
    ref = y.view(Cout, nblk, 128)[:, :, :G * K].reshape(Cout, nblk * G, K).amax(dim=2)[:, :ngout]
    refp = ops.p16_from_f32(ref.unsqueeze(0).contiguous())
    dec = ops.p16_to_f32(got)[0]
    assert torch.equal(dec[:, :ngout], ops.p16_to_f32(refp)[0])
    assert not dec[:, ngout:].any()                                  # pad columns stay zero


@pytest.mark.parametrize("C1,C2,Cout,nblk,M", [(768, 0, 1024, 4, 64), (512, 3, 768, 2, 128), (64, 0, 96, 3, 64)])
def test_gmax_f32_equals_layer_then_group_max(C1, C2, Cout, nblk, M):
    from sonet_hip import ops
    L = nblk * 128
    G = 128 // M
    g = torch.Generator().manual_seed(C1 + Cout + M)
    xp = ops.p16_from_f32(cu(torch.randn(1, C1, L, generator=g)))
    x2 = ops.p16_from_f32(cu(torch.randn(1, C2, L, generator=g))) if C2 else None
    W, sc, sh = _layer(C1, Cout, 4, C2)
    wp = ops.pointmlp_h3p_pack(W)
    ngout = nblk * G - (1 if nblk * G > 2 else 0)
    got = ops.pointmlp_h3p_gmax(xp, wp, sc, sh, False, Cout, M, G, ngout, x2=x2, out="f32")      # ngout x Cout
    y = ops.pointmlp_h3p(xp, wp, sc, sh, False, Cout, x2=x2, out="f32")[0]
    ref = y.view(Cout, nblk * G, M).amax(dim=2)[:, :ngout].t().contiguous()
    assert torch.equal(got, ref)


def test_gmax_nan_wins_and_range_is_logged():
    from sonet_hip import ops
    g = torch.Generator().manual_seed(1)
    xp = ops.p16_from_f32(cu(torch.randn(1, 64, 256, generator=g)))
    W, sc, sh = _layer(64, 64, 5)
    wp = ops.pointmlp_h3p_pack(W)
    sh2 = sh.clone()
    sh2[7] = float("nan")
    got = ops.pointmlp_h3p_gmax(xp, wp, sc, sh2, False, 64, 64, 2, 4, out="f32")
    assert torch.isnan(got[:, 7]).all() and not torch.isnan(got[:, :7]).any()
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.pointmlp_h3p_gmax(xp, wp, sc, sh, True, 64, 9, 14, 28, out="p16")
    assert rs.violations() == []
    big = sc.clone()
    big[40] = 1.0e5                                                  # one channel of a high tile leaves +-2047
    with ops.range_scope(torch.device(DEV)) as rs:
        ops.pointmlp_h3p_gmax(xp, wp, big, sh, True, 64, 9, 14, 28, out="p16")
    bad = rs.violations()
    assert bad and "2047" in bad[0][1]


@pytest.mark.parametrize("B,M,K,KI,C,avg", [(4, 64, 9, 9, 512, True), (3, 64, 9, 12, 64, False), (2, 128, 5, 5, 48, True), (1, 64, 16, 16, 32, True)])
def test_knn_stage_prepare_and_input_equal_prepare_gather_lead(B, M, K, KI, C, avg):
    """sonet_knn_stage_prepare_f32 + sonet_knn_stage_input_p16 == sonet_knn_prepare_f32 + sonet_node_gather_lead_affine_act_f32 (the round-4
    kernels) on the same (P16-representable) z, re-ordered to the node-major padded layout and split: bit for bit, incl. out-of-range
    neighbour ids (read as zeros) and the centres."""
    from sonet_hip import ops
    g = torch.Generator().manual_seed(B + M + K + C)
    coord = cu(torch.randn(B, 3, M, generator=g))
    I = torch.randint(0, M, (B, M, KI), generator=g)
    I[:, :, 0] = torch.arange(M)
    I[0, 3, 1] = M + 7                                               # ids nobody owns
    I[B - 1, M - 1, K - 1] = -1
    I = cu(I)
    Lm = ops.node_stage_columns(B, M)
    zf = torch.zeros(1, C, Lm)
    zf[0, :, :B * M] = torch.randn(C, B * M, generator=g)
    zp = ops.p16_from_f32(cu(zf))
    z = ops.p16_to_f32(zp)[0]                                        # the values the planes hold
    wl = cu(torch.randn(C, 3, generator=g))
    sc, sh = cu(torch.rand(C, generator=g) + 0.5), cu(torch.randn(C, generator=g) * 0.2)
    prep = ops.knn_stage_prepare(coord, I, K, avg)
    h1 = ops.knn_stage_input(prep, zp, wl, sc, sh, True)
    G = prep["G"]
    assert G == min(16, 128 // K)
    center, cp = prep["center"], prep["center_p16"]
    # round-4 kernels
    ctr_ref, dec, gidx = ops.knn_prepare(coord, I[:, :, :K].contiguous(), avg)
    zb = z[:, :B * M].reshape(C, B, M).permute(1, 0, 2).contiguous()
    h_ref = ops.node_gather_lead_affine_act(zb, gidx, dec, wl, sc, sh, True)       # B x C x (K M), k-major
    assert torch.equal(center, ctr_ref)
    # -> node-major, padded blocks of G nodes
    nb = (B * M + G - 1) // G
    hr = h_ref.view(B, C, K, M).permute(1, 0, 3, 2).reshape(C, B * M, K)            # C x node x k
    pad = torch.zeros(C, nb * G, K, device=DEV)
    pad[:, :B * M] = hr
    blk = torch.zeros(C, nb, 128, device=DEV)
    blk[:, :, :G * K] = pad.view(C, nb, G * K)
    ref = ops.p16_to_f32(ops.p16_from_f32(blk.view(1, C, nb * 128).contiguous()))
    assert h1.L == nb * 128
    assert torch.equal(ops.p16_to_f32(h1), ref)
    # the centre panel: channels 0..2, pad columns zero
    cdec = ops.p16_to_f32(cp)[0]
    cref = ops.p16_to_f32(ops.p16_from_f32(center.permute(1, 0, 2).reshape(1, 3, B * M).contiguous()))[0]
    assert torch.equal(cdec[:, :B * M], cref) and not cdec[:, B * M:].any()


def _opt(B, N, som_k=9, som_k_type="avg", node_num=64):
    return Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                     activation="relu", normalization="batch", dropout=0.7, node_num=node_num, k=3, som_k=som_k, som_k_type=som_k_type,
                     bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)


@pytest.mark.parametrize("B,N,som_k,ktype", [(2, 512, 9, "avg"), (3, 700, 9, "avg"), (1, 300, 5, "center"), (4, 1024, 9, "center")])
def test_encoder_node_stage_against_round4_stage_and_oracle(B, N, som_k, ktype):
    """The whole forward with the flat node-level stage == the forward with the round-4 stage (second-generation layers, planes_max,
    lastdim_max) within the f32-class bound, == the CPU oracle; the intermediate maps the reference leaves on the module
    (knn_center_1, knn_feature_1, final_pn_out) are decoded on demand and match; odd batch sizes take the padded column axis."""
    from models import networks as NW
    from oracle import cpu_oracle as O
    from sonet_hip import ops, synth
    opt = _opt(B, N, som_k, ktype)
    enc = NW.Encoder(opt)
    sd = synth.fill_state_dict_(enc.state_dict(), 21)
    cpu_sd = {k: v.clone() for k, v in sd.items()}
    enc.to(DEV).eval()
    inp = synth.make_inputs(B, N, seed=4)
    args = [inp[k].to(DEV) for k in ("pc", "sn", "node", "node_knn_I")]
    with torch.no_grad():
        assert enc._node_stage_ok(B, 64, args[3])
        f_new = enc(*args).clone()
        assert enc.__dict__["_stage"] is not None and enc.__dict__["_knn_feature_1"] is None       # nothing decoded yet
        knn_new, fin_new, ctr_new = enc.knn_feature_1.clone(), enc.final_pn_out.clone(), enc.knn_center_1.clone()
        mm_new = enc.first_pn_out_masked_max.clone()
        old = ops.NODE_STAGE_P16
        ops.NODE_STAGE_P16 = False
        try:
            f_old = enc(*args).clone()
            assert enc.__dict__["_stage"] is None
            knn_old, fin_old, ctr_old = enc.knn_feature_1.clone(), enc.final_pn_out.clone(), enc.knn_center_1.clone()
            mm_old = enc.first_pn_out_masked_max.clone()
        finally:
            ops.NODE_STAGE_P16 = old
    assert torch.equal(mm_new, mm_old) and torch.equal(ctr_new, ctr_old)
    assert knn_new.shape == knn_old.shape and fin_new.shape == fin_old.shape
    assert_close_rms(knn_new.cpu().numpy(), knn_old.cpu().numpy(), 1e-5, "knn_feature_1 vs round-4 stage")
    assert_close_rms(fin_new.cpu().numpy(), fin_old.cpu().numpy(), 1e-5, "final_pn_out vs round-4 stage")
    assert_close_rms(f_new.cpu().numpy(), f_old.cpu().numpy(), 1e-5, "feature vs round-4 stage")
    # the feature is the max of final_pn_out over the nodes
    assert_close_rms(f_new.cpu().numpy(), fin_new.amax(dim=2).cpu().numpy(), 1e-5, "feature vs max(final_pn_out)")
    ref = O.encoder_forward(cpu_sd, inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], k=opt.k, som_k=som_k, som_k_type=ktype)
    assert_close_rms(f_new.cpu().numpy(), ref["feature"].numpy(), 1e-5, "feature vs oracle")
    assert_close_rms(knn_new.cpu().numpy(), ref["knn_feature_1"].numpy(), 1e-5, "knn_feature_1 vs oracle")
    assert_close_rms(fin_new.cpu().numpy(), ref["final_pn_out"].numpy(), 1e-5, "final_pn_out vs oracle")


def test_encoder_node_stage_range_guard_recomputes():
    """A pooled value beyond the fp16-split range is caught where the split happens (the decode pass of the fused first PointNet logs it):
    the guarded forward recomputes in x3 and still matches the round-4 stage run the same way."""
    import warnings
    from models import networks as NW
    from sonet_hip import ops, synth
    B, N = 2, 512
    enc = NW.Encoder(_opt(B, N))
    sd = synth.fill_state_dict_(enc.state_dict(), 5)
    sd["first_pointnet.layers.3.conv.bias"][11] = 6000.0            # pooled channel 11 ~ 6000 > 2047
    enc.load_state_dict(sd)
    enc.to(DEV).eval()
    inp = synth.make_inputs(B, N, seed=2)
    args = [inp[k].to(DEV) for k in ("pc", "sn", "node", "node_knn_I")]
    ops._range_warned = False
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        f = enc(*args).clone()
    assert any("range" in str(x.message) for x in w)
    with torch.no_grad(), ops.precision("x3"):
        f_x3 = enc(*args).clone()
    assert torch.equal(f, f_x3)


@pytest.mark.parametrize("B,N,K,KI,avg", [(4, 700, 9, 9, True), (3, 300, 9, 12, False), (1, 5000, 5, 5, True), (2, 256, 16, 16, True)])
def test_som_stage_rider_equals_knn_stage_prepare(B, N, K, KI, avg):
    """sonet_som_assign_sort_knn_f32 (KNNModule's index / coordinate side computed by the SOM stage's second launch from the cluster means in
    LDS) == sonet_knn_stage_prepare_f32 on som_node: records (incl. padding columns), centres, centre panel -- bit for bit; the SOM outputs
    themselves are unchanged."""
    from sonet_hip import ops, synth
    M = 64
    inp = synth.make_inputs(B, N, seed=B + N)
    g = torch.Generator().manual_seed(K)
    I = torch.randint(0, M, (B, M, KI), generator=g)
    I[:, :, 0] = torch.arange(M)
    I[0, 5, K - 1] = -3
    I[B - 1, 0, 1] = M
    I = cu(I)
    x, sn, node = cu(inp["pc"]), cu(inp["sn"]), cu(inp["node"])
    a0, g0 = ops.som_assign_sort(x, sn, node, 3)
    a1, g1 = ops.som_assign_sort(x, sn, node, 3, knn=(I, K, avg))
    # (the order of a node's copies inside its run follows the first launch's LDS atomics: x_aug_sorted / pos0 are equal as multisets per node only)
    for key in ("som_node", "row_max", "ids_sorted", "node_off", "count"):
        assert torch.equal(g0[key], g1[key]), key
    assert torch.equal(a0.min_idx_i32, a1.min_idx_i32)
    ref = ops.knn_stage_prepare(g0["som_node"], I, K, avg)
    got = g1["knn_prep"]
    assert got["G"] == ref["G"] and got["Lp"] == ref["Lp"]
    assert torch.equal(got["rec"], ref["rec"])
    assert torch.equal(got["center"], ref["center"])
    assert torch.equal(got["center_p16"].data, ref["center_p16"].data)
