"""tools/microbench.py -- per-kernel timing on one MI355X (HIP events, random data, within-process A/B).

    python tools/microbench.py pointmlp        # all layer shapes of the classifier forward x MT variants
    python tools/microbench.py index_max
    python tools/microbench.py som
"""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))

import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_pointmlp(B=64):
    shapes = [(6, 0, 64, 15000), (64, 0, 128, 15000), (128, 0, 256, 15000), (64, 256, 384, 15000),
              (387, 0, 512, 576), (512, 0, 512, 576), (515, 0, 768, 64), (768, 0, 1024, 64)]
    for C1, C2, Cout, L in shapes:
        Cin = C1 + C2
        x1 = torch.randn(B, C1, L, device=DEV)
        x2 = torch.randn(B, C2, L, device=DEV) if C2 else None
        W = torch.randn(Cout, Cin, device=DEV) * (2.0 / Cin) ** 0.5
        wp = ops.pointmlp_pack(W)
        sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
        y = torch.empty(B, Cout, L, device=DEV)
        flops = 2.0 * Cin * Cout * B * L
        row = []
        for mt in (0, 2, 4, 6):
            if mt and (Cout // 32) % mt:
                continue
            for S in ((0,) if mt == 0 else (1, 2, 4)):
                if mt:
                    os.environ["SONET_POINTMLP_MT"] = str(mt)
                    os.environ["SONET_POINTMLP_S"] = str(S)
                else:
                    os.environ.pop("SONET_POINTMLP_MT", None)
                    os.environ.pop("SONET_POINTMLP_S", None)
                ms = timeit(lambda: ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y))
                row.append("MT=%s,S=%d %.3f ms %.1f TF" % (mt or "auto", S, ms, flops / ms / 1e9))
        os.environ.pop("SONET_POINTMLP_MT", None)
        os.environ.pop("SONET_POINTMLP_S", None)
        os.environ["SONET_POINTMLP_KERNEL"] = "wlds"
        ms = timeit(lambda: ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y))
        row.append("v2(wlds) auto %.3f ms %.1f TF" % (ms, flops / ms / 1e9))
        os.environ.pop("SONET_POINTMLP_KERNEL", None)
        os.environ.pop("SONET_POINTMLP_MT", None)
        os.environ.pop("SONET_POINTMLP_S", None)
        print("pointmlp %4d->%4d L=%5d B=%d : %s" % (Cin, Cout, L, B, " | ".join(row)), flush=True)


def bench_x3(B=64):
    shapes = [(6, 0, 64, 15000), (64, 0, 128, 15000), (128, 0, 256, 15000), (64, 256, 384, 15000),
              (387, 0, 512, 576), (512, 0, 512, 576), (515, 0, 768, 64), (768, 0, 1024, 64)]
    for C1, C2, Cout, L in shapes:
        Cin = C1 + C2
        x1 = torch.randn(B, C1, L, device=DEV)
        x2 = torch.randn(B, C2, L, device=DEV) if C2 else None
        W = torch.randn(Cout, Cin, device=DEV) * (2.0 / Cin) ** 0.5
        sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
        y = torch.empty(B, Cout, L, device=DEV)
        flops = 2.0 * Cin * Cout * B * L
        row = []
        wp = ops.pointmlp_pack(W, "f32")
        ms = timeit(lambda: ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y))
        row.append("f32 %.3f ms %.1f TF" % (ms, flops / ms / 1e9))
        wp3 = ops.pointmlp_pack(W, "x3")
        for mt in (0, 1, 2, 4, 6):
            if mt and (Cout // 32) % mt:
                continue
            for S in ((0,) if mt == 0 else (1, 2)):
                if mt:
                    os.environ["SONET_POINTMLP_MT"] = str(mt)
                    os.environ["SONET_POINTMLP_S"] = str(S)
                ms = timeit(lambda: ops.pointmlp(x1, wp3, sc, sh, True, Cout, x2=x2, out=y))
                row.append("x3 MT=%s,S=%d %.3f ms %.1f TF-eq" % (mt or "auto", S, ms, flops / ms / 1e9))
        os.environ.pop("SONET_POINTMLP_MT", None)
        os.environ.pop("SONET_POINTMLP_S", None)
        print("x3 %4d->%4d L=%5d B=%d : %s" % (Cin, Cout, L, B, " | ".join(row)), flush=True)


def bench_fused(B=64, L=15000):
    from models import layers as Lm
    from sonet_hip import synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    x = torch.randn(B, 6, L, device=DEV)
    flops = 2.0 * (6 * 64 + 64 * 128 + 128 * 256 + 320 * 384) * B * L
    with torch.no_grad():
        for fuse in (True, False):
            ops.FUSE_POINTRESNET = fuse
            ms = timeit(lambda: pr(x))
            print("first PointNet B=%d L=%d %s : %.3f ms  %.1f TF-eq (%.1f%% of 2500/6)" % (
                B, L, "FUSED" if fuse else "4 launches", ms, flops / ms / 1e9, flops / ms / 1e9 / (2500 / 6) * 100), flush=True)
        ops.FUSE_POINTRESNET = True
        for abl, what in ((1, "no stores"), (2, "no bf16 split"), (4, "no W streaming/barriers"), (7, "MFMA stream only")):
            os.environ["SONET_FUSED_ABLATE"] = str(abl)
            ms = timeit(lambda: pr(x))
            print("   fused ablation %-26s : %.3f ms  %.1f TF-eq" % (what, ms, flops / ms / 1e9), flush=True)
        os.environ.pop("SONET_FUSED_ABLATE", None)
        # pooled variant on node-sorted input
        from sonet_hip import synth as _s
        inp = _s.make_inputs(B, L // 3, seed=1, device=DEV)
        a = ops.som_assign(inp["pc"], inp["node"], 3)
        sg = ops.som_sort_group(inp["pc"], inp["sn"], a)
        wstream, affine = pr._fused_state()
        for abl, what in ((0, "pool full"), (8, "pool: no flush"), (16, "pool: no wave reduce"), (32, "pool: no tail publish"), (56, "pool: affine only")):
            os.environ["SONET_FUSED_ABLATE"] = str(abl)
            ms = timeit(lambda: ops.pointresnet_fused_pool(sg, wstream, affine, 64))
            print("   fused+pool %-26s : %.3f ms" % (what, ms), flush=True)
        os.environ.pop("SONET_FUSED_ABLATE", None)
        ms = timeit(lambda: ops.som_sort_group(inp["pc"], inp["sn"], a)); print("   som_sort_group %.3f ms" % ms)
        ms = timeit(lambda: ops.som_group(inp["pc"], inp["sn"], a, want_augmented=True)); print("   som_group      %.3f ms" % ms)
    ops.FUSE_POINTRESNET = True


def bench_index_max():
    for B, C, N, K in [(64, 384, 15000, 64), (8, 384, 15000, 64), (64, 384, 3072, 64)]:
        data = torch.randn(B, C, N, device=DEV)
        index = torch.randint(0, K, (B, N), device=DEV, dtype=torch.int32)
        ms = timeit(lambda: ops.index_max(data, index, K))
        byt = B * (C * N * 4 + N * 4 + C * K * 4)
        print("index_max f32 B=%d C=%d N'=%d : %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (B, C, N, ms, byt / ms / 1e6, byt / ms / 1e6 / 80), flush=True)
        d16 = data.to(torch.bfloat16)
        ms = timeit(lambda: ops.index_max(d16, index, K))
        byt = B * (C * N * 2 + N * 4 + C * K * 4)
        print("index_max bf16 B=%d C=%d N'=%d : %.4f ms  %.0f GB/s" % (B, C, N, ms, byt / ms / 1e6), flush=True)


def bench_som():
    from sonet_hip import synth
    for B, N in [(64, 5000), (8, 5000), (512, 5000)]:
        inp = synth.make_inputs(B, N, seed=1, device=DEV)
        ms_a = timeit(lambda: ops.som_assign(inp["pc"], inp["node"], 3))
        a = ops.som_assign(inp["pc"], inp["node"], 3)
        ms_g = timeit(lambda: ops.som_group(inp["pc"], inp["sn"], a, want_augmented=True))
        print("som B=%d N=%d : assign %.4f ms (%.1f Mclouds/s)  group %.4f ms" % (B, N, ms_a, B / ms_a / 1e3, ms_g), flush=True)


def bench_ablate(B=64):
    C1, C2, Cout, L = 64, 256, 384, 15000
    x1, x2 = torch.randn(B, C1, L, device=DEV), torch.randn(B, C2, L, device=DEV)
    wp = ops.pointmlp_pack(torch.randn(Cout, C1 + C2, device=DEV) * 0.08)
    sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
    y = torch.empty(B, Cout, L, device=DEV)
    flops = 2.0 * (C1 + C2) * Cout * B * L
    os.environ["SONET_POINTMLP_S"] = "2"
    for mt in (2, 6):
        os.environ["SONET_POINTMLP_MT"] = str(mt)
        for abl, what in ((0, "full"), (1, "no stores"), (2, "no X loads"), (3, "no stores, no X loads"), (4, "no LDS/barrier"), (7, "MFMA loop only")):
            os.environ["SONET_POINTMLP_ABLATE"] = str(abl)
            ms = timeit(lambda: ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y))
            print("ablate 320->384 MT=%d %-24s %.4f ms %.1f TF" % (mt, what, ms, flops / ms / 1e9), flush=True)
    os.environ.pop("SONET_POINTMLP_MT", None)
    os.environ.pop("SONET_POINTMLP_ABLATE", None)


def bench_shape_probe(B=64, L=15000):
    """Same MFMA count per workgroup, different pass structure: many cout passes x short K vs one pass x long K."""
    os.environ["SONET_POINTMLP_MT"] = "2"
    os.environ["SONET_POINTMLP_S"] = "2"
    for Cin, Cout in ((320, 384), (1920, 64), (960, 128), (160, 768), (64, 128), (256, 128), (1024, 128)):
        x1 = torch.randn(B, Cin, L if Cin <= 1024 else L // 4, device=DEV)
        Lx = x1.shape[2]
        wp = ops.pointmlp_pack(torch.randn(Cout, Cin, device=DEV) * 0.05)
        sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
        y = torch.empty(B, Cout, Lx, device=DEV)
        flops = 2.0 * Cin * Cout * B * Lx
        row = []
        for abl in (0, 3, 7):
            os.environ["SONET_POINTMLP_ABLATE"] = str(abl)
            ms = timeit(lambda: ops.pointmlp(x1, wp, sc, sh, True, Cout, out=y))
            row.append("abl=%d %.3f ms %.1f TF" % (abl, ms, flops / ms / 1e9))
        print("probe %4d->%4d L=%d : %s" % (Cin, Cout, Lx, " | ".join(row)), flush=True)
    for k in ("SONET_POINTMLP_MT", "SONET_POINTMLP_S", "SONET_POINTMLP_ABLATE"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pointmlp", "index_max", "som"]
    if "pointmlp" in which:
        bench_pointmlp()
    if "index_max" in which:
        bench_index_max()
    if "som" in which:
        bench_som()
    if "ablate" in which:
        bench_ablate()
    if "x3" in which:
        bench_x3()
    if "fused" in which:
        bench_fused()
    if "probe" in which:
        bench_shape_probe()
