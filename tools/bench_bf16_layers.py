"""tools/bench_bf16_layers.py -- the bf16 per-layer kernel (pointmlp_bf16.hip) on the shapes the bf16 training step launches at
B = 64 (forward layers, the dgrads = the same kernel on the transposed pack, with / without the statistics epilogue), against the
HBM roofline: bytes = (Cin + Cout) x B x L x 2, each once.

  python tools/bench_bf16_layers.py [--batch 64]
  SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so SONET_BF16_NXB=2 python tools/bench_bf16_layers.py   (knobs: variants build)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))

import torch  # noqa: E402

from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
# (name, C1, C2, Cout, L, stats)
SHAPES = [("fwd 6->64 +stats", 6, 0, 64, 15000, True),
          ("fwd 64->128 +stats", 64, 0, 128, 15000, True),
          ("fwd 128->256 +stats", 128, 0, 256, 15000, True),
          ("fwd 64->128", 64, 0, 128, 15000, False),
          ("fwd 128->256", 128, 0, 256, 15000, False),
          ("fwd 320->384", 64, 256, 384, 15000, False),
          ("dgrad 256->128", 256, 0, 128, 15000, False),
          ("dgrad 128->64", 128, 0, 64, 15000, False),
          ("seg 1024->512", 1024, 0, 512, 3072, False),
          ("knn 512->512", 512, 0, 512, 576, False),
          ("final 768->1024", 768, 0, 1024, 64, False)]


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=0, help="override the 15000 columns per cloud (row alignment experiments)")
    args = ap.parse_args()
    B = args.batch
    g = torch.Generator().manual_seed(0)
    tot = 0.0
    for name, C1, C2, Cout, L, stats in SHAPES:
        if args.points and L == 15000:
            L = args.points
        Cin = C1 + C2
        x1 = torch.randn(B, C1, L, generator=g).to(DEV).bfloat16()
        x2 = torch.randn(B, C2, L, generator=g).to(DEV).bfloat16() if C2 else None
        W = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
        wp = ops.pointmlp_pack(W, "bf16")
        y = torch.empty(B, Cout, L, dtype=torch.bfloat16, device=DEV)
        if stats:
            fn = lambda: ops.pointmlp_stats(x1, wp, sc, sh, False, Cout, x2=x2)       # noqa: E731
        else:
            fn = lambda: ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y)       # noqa: E731
        timeit(fn, iters=30)
        ms = timeit(fn)
        byt = 2.0 * (Cin + Cout) * B * L
        flop = 2.0 * Cin * Cout * B * L
        tot += ms
        print("%-22s (L=%5d): %8.4f ms  %6.2f TB/s (%.2f of 8)  %7.1f TFLOP/s" % (name, L, ms, byt / ms / 1e9, byt / ms / 8e9, flop / ms / 1e9), flush=True)
        del x1, x2, y
        torch.cuda.empty_cache()
    print("sum %.4f ms" % tot)


if __name__ == "__main__":
    main()
