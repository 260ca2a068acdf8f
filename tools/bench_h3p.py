"""tools/bench_h3p.py -- the third-generation fp16-split layer (pointmlp_h3p.hip, P16 inputs) against the second generation
(pointmlp_x3.hip: f32 inputs, split in the kernel) on the layer shapes the models launch at B = 64.

  SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so python tools/bench_h3p.py [--sweep] [--shapes big|node|all]

--sweep (variants build): every tile shape (SONET_H3P_SHAPE = MT,NC,OCC) and slab count the shape allows.
Prints one line per (layer, variant): mean ms over the timed launches (HIP events on the launching stream), TFLOP/s-equivalent
(2 Cin Cout B L / t; x 3 MFMA terms = matrix-pipe work) and the algorithmic HBM rate (inputs + outputs once)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))

import torch  # noqa: E402

from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"

# (name, C1, C2, Cout, L, nodeadd)
BIG = [("first-pn 320->384", 64, 256, 384, 15000, False),
       ("first-pn 128->256", 128, 0, 256, 15000, False),
       ("first-pn 64->128", 64, 0, 128, 15000, False),
       ("seg 393->1024 +node", 384, 9, 1024, 3072, True),
       ("seg 1024->512", 1024, 0, 512, 3072, False),
       ("seg 512->256", 512, 0, 256, 3072, False)]
NODE = [("knn 512->512", 512, 0, 512, 576, False),
        ("knn 387->512", 384, 3, 512, 576, False),
        ("final 515->768", 512, 3, 768, 64, False),
        ("final 768->1024", 768, 0, 1024, 64, False)]


def timeit(fn, iters=20, warm=3):
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


def line(tag, ms, flop, bytes_):
    print("    %-34s %8.4f ms  %7.1f TF-eq  %6.2f TB/s" % (tag, ms, flop / ms / 1e9, bytes_ / ms / 1e9), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    B = args.batch
    shapes = (BIG if args.shapes in ("big", "all") else []) + (NODE if args.shapes in ("node", "all") else [])
    g = torch.Generator().manual_seed(0)
    for name, C1, C2, Cout, L, nodeadd in shapes:
        Cin = C1 + C2
        x1 = torch.randn(B, C1, L, generator=g).to(DEV)
        x2 = torch.randn(B, C2, L, generator=g).to(DEV) if C2 else None
        W = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
        z = torch.randn(B, Cout, 64, generator=g).to(DEV) if nodeadd else None
        zi = torch.randint(0, 64, (B, L), generator=g, dtype=torch.int32).to(DEV) if nodeadd else None
        flop = 2.0 * Cin * Cout * B * L
        byt = 4.0 * (Cin + Cout) * B * L
        print("%s  (B=%d, L=%d): %.1f GFLOP, %.2f GB in+out" % (name, B, L, flop / 1e9, byt / 1e9), flush=True)
        wp2 = ops.pointmlp_pack(W, "h3")
        gen2 = ((lambda: ops.pointmlp_nodeadd(x1, wp2, sc, sh, True, Cout, z, zi, x2=x2)) if nodeadd
                else (lambda: ops.pointmlp(x1, wp2, sc, sh, True, Cout, x2=x2)))
        timeit(gen2, iters=40)                                 # (the first launches behind the host-side set-up run ~15 % slow: clocks)
        line("2nd gen (f32 in, f32 out)", timeit(gen2), flop, byt)
        line("p16_from_f32 (both inputs)", timeit(lambda: (ops.p16_from_f32(x1), ops.p16_from_f32(x2) if C2 else None)), 0.0, 8.0 * Cin * B * L)
        p1, p2 = ops.p16_from_f32(x1), (ops.p16_from_f32(x2) if C2 else None)
        wp = ops.pointmlp_h3p_pack(W)
        kw = dict(x2=p2, z=z, zidx=zi)
        variants = [("default", None, None)]
        if args.sweep:
            CT = Cout // 32
            for sh_ in ("12,1,1", "8,2,1", "6,2,1", "4,2,2", "4,1,2", "2,1,2"):
                mt = int(sh_.split(",")[0])
                if CT % mt:
                    continue
                groups = CT // mt
                if sh_ == "12,1,1" and nodeadd:
                    continue
                for ns in sorted({1, groups} | ({2} if groups % 2 == 0 else set()) | ({4} if groups % 4 == 0 else set())):
                    if CT // ns > 32:
                        continue
                    variants.append(("shape %s slabs %d" % (sh_, ns), sh_, ns))
        for tag, shape, ns in variants:
            for k in ("SONET_H3P_SHAPE", "SONET_H3P_NSLAB"):
                os.environ.pop(k, None)
            if shape:
                os.environ["SONET_H3P_SHAPE"] = shape
                os.environ["SONET_H3P_NSLAB"] = str(ns)
            try:
                line("3rd gen %s -> f32" % tag, timeit(lambda: ops.pointmlp_h3p(p1, wp, sc, sh, True, Cout, out="f32", **kw)), flop, byt)
                line("3rd gen %s -> p16" % tag, timeit(lambda: ops.pointmlp_h3p(p1, wp, sc, sh, True, Cout, out="p16", **kw)), flop, byt)
            except Exception as e:                            # a shape the launcher refuses
                print("    %-34s %s" % (tag, str(e)[:90]))
        line("2nd gen again (after the others)", timeit(gen2), flop, byt)
        del x1, x2, p1, p2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
