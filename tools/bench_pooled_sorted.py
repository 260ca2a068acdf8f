"""tools/bench_pooled_sorted.py -- the sparse backward of the pooled last layer (pooled_dgrad / pooled_wgrad, f32) on the entries of a REAL
training step (B = 64 x 5000 synthetic points of bench.py): positions in the original column order (store + index_max path) and in the
node-sorted order of the f32-class training path, stand-alone and side by side on two streams.  With the variants library
(SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so) also the ablations of the dgrad kernel (SONET_PD_ABL: 1 no stores, 2 no accumulation,
4 no sort)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, N = 64, 5000
cap = {}
orig_d, orig_w = ops.pooled_dgrad, ops.pooled_wgrad


def grab_d(g_pooled, pos, w, C1, C2, L, out_dtype=torch.float32, wt_pack=None, **tail):
    cap["d"] = (g_pooled.clone(), pos.clone(), w.clone(), C1, C2, L)
    return orig_d(g_pooled, pos, w, C1, C2, L, out_dtype=out_dtype, wt_pack=wt_pack, **tail)


def grab_w(g_t, pos_t, x, xaff=None):
    cap.setdefault("w", []).append((g_t.clone(), pos_t.clone(), x.clone(), xaff))
    return orig_w(g_t, pos_t, x, xaff)


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


variants_lib = "variants" in os.path.basename(_lib.LIB_PATH)
for tag, seg in (("original column order", False), ("node-sorted columns", True)):
    cap.clear()
    ops.H3_SEGPOOL, ops.H3_NORM_ON_LOAD = seg, False
    ops.pooled_dgrad, ops.pooled_wgrad = grab_d, grab_w
    with ops.precision("h3"):
        opt = bench.make_opt(dev, B, N)
        enc, cls = NW.Encoder(opt), NW.Classifier(opt)
        enc.want_first_pn_out = False
        synth.fill_state_dict_(enc.state_dict(), 0)
        synth.fill_state_dict_(cls.state_dict(), 1)
        enc.to(dev).train()
        cls.to(dev).train()
        inp = synth.make_inputs(B, N, seed=100, device=dev)
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
        torch.nn.functional.cross_entropy(cls(feat, 0), inp["label"]).backward()
        cnt = enc._lazy["a"].count
    ops.pooled_dgrad, ops.pooled_wgrad = orig_d, orig_w
    del enc, cls, feat
    g_pooled, pos, w, C1, C2, L = cap["d"]
    E = pos.shape[1] * pos.shape[2]
    p = torch.where(pos < 0, L, pos).long().reshape(B, -1)
    per_col = torch.zeros(B, L + 1, device=dev).scatter_add_(1, p, torch.ones(B, E, device=dev))[:, :L]
    per_bucket = per_col[:, : (L // 32) * 32].reshape(B, -1, 32).sum(2)
    per_tile = per_col[:, : (L // 128) * 128].reshape(B, -1, 128).sum(2)
    print("== %s: node sizes %d .. %d; entries per column: max %d, columns with none %.0f %%; per 32-column bucket: mean %.0f max %d, sum of squares / (mean^2 n) = %.2f; per 128-column tile max %d"
          % (tag, int(cnt.min()), int(cnt.max()), int(per_col.max()), 100.0 * float((per_col == 0).float().mean()), float(per_bucket.mean()), int(per_bucket.max()),
             float((per_bucket ** 2).mean() / per_bucket.mean() ** 2), int(per_tile.max())))
    t_d = timeit(lambda: ops.pooled_dgrad(g_pooled, pos, w, C1, C2, L))
    ws = cap["w"]
    t_w = [timeit(lambda q=q: ops.pooled_wgrad(q[0], q[1], q[2], q[3])) for q in ws]
    print("   pooled_dgrad f32 %.4f ms   pooled_wgrad x1 / x2 %.4f / %.4f ms" % (t_d, t_w[0], t_w[1]))
    side = torch.cuda.Stream(device=dev)

    def both():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for q in ws:
                ops.pooled_wgrad(q[0], q[1], q[2], q[3])
        ops.pooled_dgrad(g_pooled, pos, w, C1, C2, L)
        torch.cuda.current_stream().wait_stream(side)
    print("   side by side on two streams: %.4f ms" % timeit(both))
    if seg and C2 and (C1 + C2) % 4 == 0:
        # what follows the launch in the training step, as launches of their own against riding on its store (ops.POOLED_DGRAD_TAIL)
        gen = torch.Generator().manual_seed(5)
        raw2 = torch.randn(B, C2, L, generator=gen).to(dev)
        sc2, sh2 = (torch.rand(C2, generator=gen) + 0.5).to(dev), (torch.randn(C2, generator=gen) * 0.3).to(dev)
        col0 = (torch.randn(B, C1 + C2, generator=gen) * 1e-4).to(dev)
        p0 = torch.randint(0, L, (B,), generator=gen, dtype=torch.int32).to(dev)
        p0l = p0.long().view(B, 1, 1)

        def apart():
            a, b2 = ops.pooled_dgrad(g_pooled, pos, w, C1, C2, L)
            a.scatter_add_(2, p0l.expand(B, C1, 1), col0[:, :C1].unsqueeze(2))
            b2.scatter_add_(2, p0l.expand(B, C2, 1), col0[:, C1:].unsqueeze(2))
            return ops.pointwise_bwd_stats(b2, raw2, sc2, sh2, True, want_sums=True)
        t_a = timeit(apart)
        t_b = timeit(lambda: ops.pooled_dgrad(g_pooled, pos, w, C1, C2, L, col0=col0, pos0=p0, below=(raw2, sc2, sh2, True)))
        t_c = timeit(lambda: ops.pooled_dgrad(g_pooled, pos, w, C1, C2, L, col0=col0, pos0=p0))
        print("   pooled_dgrad + 2 scatter_add + statistics pass of the %d-channel layer: %.4f ms   on the store of the launch: %.4f ms (column 0 only: %.4f ms)"
              % (C2, t_a, t_b, t_c))
    if variants_lib:
        for abl, what in ((4, "no sort"), (2, "no accumulation"), (1, "no stores"), (6, "no sort, no accumulation")):
            os.environ["SONET_PD_ABL"] = str(abl)
            print("   pooled_dgrad, %-26s %.4f ms" % (what + ":", timeit(lambda: ops.pooled_dgrad(g_pooled, pos, w, C1, C2, L))))
        os.environ.pop("SONET_PD_ABL", None)
