"""tools/bench_node_stage.py -- the launches of the flat node-level stage (csrc/node_stage.hip, the group-max epilogues of pointmlp_h3p.hip) at
B = 64, M = 64, K = 9, one by one, warm, HIP events around back-to-back launches; with the variants library also over tile shapes / slab counts.

  [SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so] python tools/bench_node_stage.py [--sweep]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))

import torch  # noqa: E402

from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=30, warm=5):
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
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    B, M, K = args.batch, 64, 9
    g = torch.Generator().manual_seed(0)
    Lm = ops.node_stage_columns(B, M)
    coord = torch.randn(B, 3, M, generator=g).to(DEV)
    I = torch.randint(0, M, (B, M, K), generator=g).to(DEV)

    def lyr(cin, cout):
        W = (torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5).to(DEV)
        return ops.pointmlp_h3p_pack(W), (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.1).to(DEV)

    xp = ops.p16_from_f32(torch.randn(1, 384, Lm, generator=g).to(DEV))
    w1, s1, t1 = lyr(384, 512)
    w2, s2, t2 = lyr(512, 512)
    w3, s3, t3 = lyr(515, 768)
    w4, s4, t4 = lyr(768, 1024)
    wl = torch.randn(512, 3, generator=g).to(DEV)
    prep = ops.knn_stage_prepare(coord, I, K, True)
    z = ops.pointmlp_h3p(xp, w1, s1, t1, False, 512, out="p16")
    h1 = ops.knn_stage_input(prep, z, wl, s1, t1, True)
    knn = ops.pointmlp_h3p_gmax(h1, w2, s2, t2, True, 512, K, prep["G"], B * M, out="p16", Lout=Lm)
    h3 = ops.pointmlp_h3p(knn, w3, s3, t3, True, 768, x2=prep["center_p16"], out="p16")
    cases = [
        ("knn_stage_prepare", 0.0, lambda: ops.knn_stage_prepare(coord, I, K, True), False),
        ("z 384->512 L%d -> p16" % Lm, 2.0 * 384 * 512 * Lm, lambda: ops.pointmlp_h3p(xp, w1, s1, t1, False, 512, out="p16"), True),
        ("knn_stage_input 512 x %d" % prep["Lp"], 0.0, lambda: ops.knn_stage_input(prep, z, wl, s1, t1, True), False),
        ("knn 512->512 gmax9 L%d" % prep["Lp"], 2.0 * 512 * 512 * B * M * K,
         lambda: ops.pointmlp_h3p_gmax(h1, w2, s2, t2, True, 512, K, prep["G"], B * M, out="p16", Lout=Lm), True),
        ("knn 512->512 plain -> p16 (no max)", 2.0 * 512 * 512 * B * M * K, lambda: ops.pointmlp_h3p(h1, w2, s2, t2, True, 512, out="p16"), True),
        ("final 515->768 L%d -> p16" % Lm, 2.0 * 515 * 768 * Lm, lambda: ops.pointmlp_h3p(knn, w3, s3, t3, True, 768, x2=prep["center_p16"], out="p16"), True),
        ("final 768->1024 gmax64 L%d" % Lm, 2.0 * 768 * 1024 * Lm, lambda: ops.pointmlp_h3p_gmax(h3, w4, s4, t4, False, 1024, M, 128 // M, B, out="f32"), True),
        ("final 768->1024 plain -> f32 (no max)", 2.0 * 768 * 1024 * Lm, lambda: ops.pointmlp_h3p(h3, w4, s4, t4, False, 1024, out="f32"), True),
    ]
    for name, flop, fn, layer in cases:
        variants = [("default", None, None)]
        if args.sweep and layer:
            cout = 512 if "512->512" in name or "384" in name else 768 if "768 L" in name else 1024
            CT = cout // 32
            for sh_ in ("4,1,2", "2,1,2", "4,2,2", "8,2,1"):
                if "gmax" in name and sh_.split(",")[1] != "1":
                    continue
                mt = int(sh_.split(",")[0])
                if CT % mt:
                    continue
                groups = CT // mt
                for ns in sorted({groups} | ({groups // 2} if groups % 2 == 0 else set())):
                    variants.append(("shape %s slabs %d" % (sh_, ns), sh_, ns))
        for tag, shape, ns in variants:
            for k in ("SONET_H3P_SHAPE", "SONET_H3P_NSLAB"):
                os.environ.pop(k, None)
            if shape:
                os.environ["SONET_H3P_SHAPE"] = shape
                os.environ["SONET_H3P_NSLAB"] = str(ns)
            try:
                ms = timeit(fn)
                print("%-44s %-22s %8.4f ms %8.1f TF-eq" % (name, tag, ms, flop / ms / 1e9), flush=True)
            except Exception as e:
                print("%-44s %-22s %s" % (name, tag, str(e)[:80]))
    for k in ("SONET_H3P_SHAPE", "SONET_H3P_NSLAB"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
