#!/usr/bin/env python3
"""Bound-then-refine arg-max for the 320 -> 384 layer of the first PointNet: what would it cost?  (CPU study, no GPU, no oracle.)

VERDICT r05 item 1 / DESIGN section 9: the fused kernel spends 1440 of its 1944 MFMAs per 64 points on the last layer
(models/layers.py:431), whose 384 x kN outputs per cloud exist only to be max-pooled per (channel, node)
(models/networks.py:180-185).  Idea: ONE fp16 product per output (a third of the matrix work) bounds every output to within a
rigorous eps(channel, column); only columns whose upper bound reaches their bin's best lower bound need the three-term product.

This script measures, on the synthetic ModelNet40-shaped batch of bench.py (seeded weights, node-sorted columns), the quantities that
decide whether that pays on the MFMA pipe:

  * is the bound rigorous (|three-term - one-term| <= eps on every element) and how tight;
  * candidates per (channel, node) bin under three rules for the threshold L a column is compared with:
      final   -- L = the bin's best lower bound over ALL its columns (needs a second sweep or node-aligned workgroups),
      seq     -- L = best lower bound over the 32-point tiles processed so far in column order + the tile's own (one wave after another),
      tile    -- L = the 32-point tile's own best lower bound only (what concurrent waves / workgroups can see without communication);
  * what the matrix cores must then do: the refinement is an MFMA on a 32-point tile x 32 GATHERED channels (the activations of the
    wave's 64 points live in its registers, so only the weight fragment can be gathered).  G = ceil(distinct candidate channels of
    a 32-point tile / 32) gathered tiles, per 96-channel pass (as the kernel's accumulators are organised) and over all 384 channels.

Output: a table + the MFMA count per 64 points of the kernel under each rule, against today's 1944.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
from sonet_hip import synth  # noqa: E402


def first_pointnet_input_of_last_layer(B, N, k, M, seed):
    inp = synth.make_inputs(B, N, M=M, seed=seed)
    pc, sn, node = inp["pc"], inp["sn"], inp["node"]
    d = ((pc.unsqueeze(3) - node.unsqueeze(2)) ** 2).sum(dim=1)                 # B x N x M
    _, idx = torch.topk(d, k=k, dim=2, largest=False, sorted=True)             # B x N x k
    min_idx = idx.permute(0, 2, 1).reshape(B, k * N)                           # k-major (util/som.py:261-266)
    xs = torch.cat([pc] * k, dim=2)
    sns = torch.cat([sn] * k, dim=2)
    onehot = F.one_hot(min_idx, M).float()                                     # B x kN x M
    cnt = onehot.sum(1)                                                        # B x M
    som_node = torch.einsum("bcl,blm->bcm", xs, onehot) / (cnt.unsqueeze(1) + 1e-5)
    centers = torch.gather(som_node, 2, min_idx.unsqueeze(1).expand(B, 3, k * N))
    x_aug = torch.cat((xs - centers, sns), dim=1)                              # B x 6 x kN
    sd = {}
    dims = [6, 64, 128, 256]
    for i in range(3):
        p = "first_pointnet.layers.%d." % i
        sd[p + "conv.weight"] = torch.zeros(dims[i + 1], dims[i], 1)
        sd[p + "conv.bias"] = torch.zeros(dims[i + 1])
        for q in ("weight", "bias", "running_mean", "running_var"):
            sd[p + "norm." + q] = torch.zeros(dims[i + 1])
    sd["first_pointnet.layers.3.conv.weight"] = torch.zeros(384, 320, 1)
    sd["first_pointnet.layers.3.conv.bias"] = torch.zeros(384)
    synth.fill_state_dict_(sd, 7)

    def layer(i, x):
        p = "first_pointnet.layers.%d." % i
        y = F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"])
        y = F.batch_norm(y, sd[p + "norm.running_mean"], sd[p + "norm.running_var"], sd[p + "norm.weight"], sd[p + "norm.bias"], False, 0.1, 1e-5)
        return F.relu(y)
    l0 = layer(0, x_aug)
    t = layer(2, layer(1, l0))
    x4 = torch.cat((l0, t), dim=1)                                             # B x 320 x kN
    order = torch.argsort(min_idx, dim=1, stable=True)                         # node-sorted columns (the pooled path's layout)
    x4 = torch.gather(x4, 2, order.unsqueeze(1).expand(B, 320, k * N))
    ids = torch.gather(min_idx, 1, order)
    return x4.numpy(), ids.numpy(), sd["first_pointnet.layers.3.conv.weight"][:, :, 0].numpy()


def split16(a):
    h = a.astype(np.float16).astype(np.float32)
    m = (a - h).astype(np.float16).astype(np.float32)
    return h, m


def study(B, N, k, M, seed):
    x4, ids, W = first_pointnet_input_of_last_layer(B, N, k, M, seed)
    wh, wm = split16(W)
    nwh, nwm, nw = np.linalg.norm(wh, axis=1), np.linalg.norm(wm, axis=1), np.linalg.norm(W, axis=1)
    rows = []
    for b in range(B):
        x = x4[b]
        xh, xm = split16(x)
        L = x.shape[1]
        one = (wh.astype(np.float64) @ xh.astype(np.float64))
        three = one + wh.astype(np.float64) @ xm.astype(np.float64) + wm.astype(np.float64) @ xh.astype(np.float64)
        nxh, nxm, nx = np.linalg.norm(xh, axis=0), np.linalg.norm(xm, axis=0), np.linalg.norm(x, axis=0)
        # Cauchy-Schwarz on the two dropped terms + 2^-20 |w||x| for the f32 accumulation of 320 products on either side
        eps = np.outer(nwh, nxm) + np.outer(nwm, nxh) + 2.0 ** -20 * np.outer(nw, nx)
        diff = np.abs(three - one)
        assert (diff <= eps).all(), "the bound is not rigorous"
        up, lo = one + eps, one - eps
        node = ids[b]
        tile32 = np.arange(L) // 32
        ntile = int(tile32[-1]) + 1
        # --- rule 'final'
        Lfin = np.full((384, M), -np.inf)
        for m in range(M):
            sel = node == m
            if sel.any():
                Lfin[:, m] = lo[:, sel].max(axis=1)
        cand_final = up >= Lfin[:, node]
        # --- rules 'seq' and 'tile': walk the 32-point tiles in order
        cand_seq = np.zeros_like(cand_final)
        cand_tile = np.zeros_like(cand_final)
        Lrun = np.full((384, M), -np.inf)
        for t in range(ntile):
            c0, c1 = 32 * t, min(32 * t + 32, L)
            nd = node[c0:c1]
            Lt = np.full((384, M), -np.inf)
            for m in np.unique(nd):
                Lt[:, m] = lo[:, c0:c1][:, nd == m].max(axis=1)
            cand_tile[:, c0:c1] = up[:, c0:c1] >= Lt[:, nd]
            Lrun = np.maximum(Lrun, Lt)
            cand_seq[:, c0:c1] = up[:, c0:c1] >= Lrun[:, nd]
        bins = int((np.bincount(node, minlength=M) > 0).sum()) * 384

        def gathered(cand):
            g_all, g_pass, chans = [], [], []
            for t in range(ntile):
                ch = cand[:, 32 * t:32 * t + 32].any(axis=1)
                chans.append(int(ch.sum()))
                g_all.append(math.ceil(ch.sum() / 32))
                g_pass.append(sum(math.ceil(ch[96 * p:96 * p + 96].sum() / 32) for p in range(4)))
            return np.array(chans), np.array(g_all), np.array(g_pass)
        for name, cand in (("final", cand_final), ("seq", cand_seq), ("tile", cand_tile)):
            chans, g_all, g_pass = gathered(cand)
            rows.append(dict(cloud=b, rule=name, cand_per_bin=cand.sum() / bins, cand_per_column=cand.sum() / L,
                             columns_with_a_candidate=float(cand.any(axis=0).mean()),
                             tiles32x32_with_a_candidate=float(np.mean([[cand[32 * ct:32 * ct + 32, 32 * t:32 * t + 32].any() for ct in range(12)] for t in range(ntile)])),
                             distinct_channels_per_32pt_tile=float(chans.mean()), G_all=float(g_all.mean()), G_pass_sum=float(g_pass.mean())))
        rows.append(dict(cloud=b, rule="bound", eps_over_rms_out=float(eps.mean() / three.std()), diff_over_eps_max=float((diff / eps).max()),
                         diff_over_eps_mean=float((diff / eps).mean())))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=2)
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rows = study(a.clouds, a.points, 3, 64, a.seed)
    for r in rows:
        print("  ".join("%s=%s" % (k, ("%.4g" % v) if isinstance(v, float) else v) for k, v in r.items()))
    # MFMA count per 64 points (two 32-point tiles) of the fused kernel: layers 1-3 + second layer-1 run = 504; last layer today 1440
    print("\nMFMAs per 64 points and wave (today 1944 = 504 + 1440):")
    for rule in ("final", "seq", "tile"):
        rr = [r for r in rows if r["rule"] == rule]
        g_all = np.mean([r["G_all"] for r in rr])
        g_pass = np.mean([r["G_pass_sum"] for r in rr])
        # one-term sweep: 20 chunks x 12 tiles x 2 column tiles = 480.  Refinement per 32-point tile and gathered tile: 20 chunks x 3 terms
        # (recompute, bit-identical chain) or x 2 terms (correction added to the one-term value: a different, equally rigorous arithmetic)
        for terms in (3, 2):
            after_all = 504 + 480 + 2 * g_all * 20 * terms
            per_pass = 504 + 480 + 2 * g_pass * 20 * terms
            print("  rule %-5s refinement %d-term: gathered after all passes %.0f (%.2f x)   gathered per 96-channel pass %.0f (%.2f x)"
                  % (rule, terms, after_all, after_all / 1944, per_pass, per_pass / 1944))


if __name__ == "__main__":
    main()
