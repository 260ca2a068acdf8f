"""tools/check_sorted_bwd.py -- the sparse pooled dgrad / wgrad kernels on the positions the sorted pool hands them, against float64 torch (scatter + matmul)."""
import os, sys
from argparse import Namespace
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd")); sys.path.insert(0, ROOT)
import torch
from models import networks as NW
from sonet_hip import ops, synth
DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", (sys.argv[1] if len(sys.argv) > 1 else "train_step_b8_n5000") + ".npz"))
B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).contiguous()
orig_d, orig_w = ops.pooled_dgrad, ops.pooled_wgrad

def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())

def dgrad(g_pooled, pos, w, C1, C2, L, out_dtype=torch.float32, wt_pack=None):
    r = orig_d(g_pooled, pos, w, C1, C2, L, out_dtype=out_dtype, wt_pack=wt_pack)
    Bq, C, M = g_pooled.shape
    G = torch.zeros(Bq, C, L + 1, dtype=torch.float64, device=g_pooled.device).scatter_add_(2, torch.where(pos < 0, L, pos).long(), g_pooled.double())[:, :, :L]
    ref = torch.matmul(w.double().t().unsqueeze(0), G)
    got = torch.cat([r[0]] + ([r[1]] if C2 else []), dim=1)
    cnt = torch.zeros(Bq, L + 1, device=pos.device).scatter_add_(1, torch.where(pos < 0, L, pos).long().reshape(Bq, -1), torch.ones(Bq, C * M, device=pos.device))[:, :L]
    bucket = cnt.reshape(Bq, -1)[:, : (L // 32) * 32].reshape(Bq, -1, 32).sum(2)
    print("  pooled_dgrad: max err / max |ref| = %.3g   (entries per column max %d, per 32-column bucket max %d)" % (rel(got, ref), int(cnt.max()), int(bucket.max())))
    return r

def wgrad(g_t, pos_t, x, xaff=None):
    r = orig_w(g_t, pos_t, x, xaff)
    xx = x.double()
    if xaff is not None:
        xx = xx * xaff[0].double().view(1, -1, 1) + xaff[1].double().view(1, -1, 1)
        if xaff[2]:
            xx = xx.clamp_min(0)
    Bq, M, C = g_t.shape
    gath = torch.gather(xx, 2, pos_t.reshape(Bq, 1, M * C).long().clamp_min(0).expand(Bq, xx.shape[1], M * C))     # B x Ci x (M C)
    ref = torch.einsum("bie,be->ei", gath, g_t.reshape(Bq, M * C).double()).reshape(M, C, -1).sum(0)
    print("  pooled_wgrad (Ci=%d): max err / max |ref| = %.3g" % (x.shape[1], rel(r, ref)))
    return r

ops.pooled_dgrad, ops.pooled_wgrad = dgrad, wgrad
for tag, seg, nol in (("A_store", False, False), ("B_sorted", True, False), ("C_sorted_nol", True, True)):
    ops.H3_SEGPOOL, ops.H3_NORM_ON_LOAD = seg, nol
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    enc.want_first_pn_out = False
    synth.fill_state_dict_(enc.state_dict(), seed); synth.fill_state_dict_(cls.state_dict(), seed + 1)
    enc.to(DEV).train(); cls.to(DEV).train()
    print(tag)
    with ops.precision("h3"):
        feat = enc(cu(g["pc"]), cu(g["sn"]), cu(g["node"]), cu(g["node_knn_I"]), is_train=True, epoch=0)
        a = enc._lazy["a"]
        print("  node counts: min %d max %d, empty nodes %d" % (int(a.count.min()), int(a.count.max()), int((a.count == 0).sum())))
        loss = torch.nn.functional.cross_entropy(cls(feat, 0), cu(g["label"]))
        loss.backward()
