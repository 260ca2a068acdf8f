"""tools/bench_mall.py -- does the 256 MiB Infinity Cache serve the second pass of a two-pass element-wise pair?  The BatchNorm / ReLU backward
reads (gy, raw) twice: statistics, then apply.  Per channel slab (the statistics are per channel) the pair's working set can be made
smaller than the cache.  Measured: the apply pass cold (after a 1 GiB flush), and right after the statistics pass on the same slab.

  python tools/bench_mall.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV, B, L = "cuda:0", 64, 15000
flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV)              # 1 GiB


def cold():
    flush.add_(1.0)


def ev():
    return torch.cuda.Event(enable_timing=True)


for dt in (torch.bfloat16, torch.float32):
    eb = 2 if dt == torch.bfloat16 else 4
    for C in (4, 8, 16, 32, 64):
        g = torch.Generator().manual_seed(C)
        gy = torch.randn(B, C, L, generator=g).to(dt).to(DEV)
        raw = torch.randn(B, C, L, generator=g).to(dt).to(DEV)
        sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
        a, b, c0 = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        n = B * C * L
        res = {"apply cold": [], "stats cold": [], "apply after stats": [], "apply after apply": []}
        for _ in range(6):
            cold()
            e = [ev() for _ in range(6)]
            e[0].record()
            ops.pointwise_bwd_stats(gy, raw, sc, sh, True)
            e[1].record()
            ops.pointwise_bwd_apply(gy, raw, sc, sh, True, a, b, c0)
            e[2].record()
            ops.pointwise_bwd_apply(gy, raw, sc, sh, True, a, b, c0)
            e[3].record()
            cold()
            e[4].record()
            ops.pointwise_bwd_apply(gy, raw, sc, sh, True, a, b, c0)
            e[5].record()
            torch.cuda.synchronize()
            res["stats cold"].append(e[0].elapsed_time(e[1]))
            res["apply after stats"].append(e[1].elapsed_time(e[2]))
            res["apply after apply"].append(e[2].elapsed_time(e[3]))
            res["apply cold"].append(e[4].elapsed_time(e[5]))
        med = {k: sorted(v[1:])[len(v[1:]) // 2] for k, v in res.items()}
        print("%-8s C=%2d (gy + raw = %5.1f MB): stats cold %.4f ms (%.2f TB/s) | apply cold %.4f (%.2f TB/s)  after stats %.4f (%.2f TB/s)  after apply %.4f (%.2f TB/s)" % (
            str(dt).split(".")[-1], C, 2 * n * eb / 1e6, med["stats cold"], 2 * n * eb / med["stats cold"] / 1e9,
            med["apply cold"], 3 * n * eb / med["apply cold"] / 1e9, med["apply after stats"], 3 * n * eb / med["apply after stats"] / 1e9,
            med["apply after apply"], 3 * n * eb / med["apply after apply"] / 1e9), flush=True)
        del gy, raw
        torch.cuda.empty_cache()
