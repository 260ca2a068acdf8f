"""tools/bench_h3r.py -- the fp16-split layer kernel, second generation (pointmlp_h3r_kernel) against the first
(SONET_POINTMLP_H3R=0): bit-identical outputs and time per launch at the node-level / head / training shapes."""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # (B, C1, C2, Cout, L, gather L1 or 0)
    (64, 515, 0, 768, 64, 0), (64, 768, 0, 1024, 64, 0), (64, 387, 0, 512, 576, 0), (64, 512, 0, 512, 576, 0),
    (64, 384, 3, 512, 576, 64), (64, 320, 0, 384, 15000, 0), (64, 128, 0, 256, 15000, 0), (64, 64, 0, 128, 15000, 0),
    (64, 256, 64, 384, 15000, 0), (8, 1347, 0, 512, 1024, 0), (3, 40, 0, 128, 77, 0), (2, 17, 0, 128, 1, 0),
]


def run(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B, C1, C2, Cout, L, L1 in SHAPES:
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = (torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(DEV)
    x1 = torch.randn(B, C1, L1 if L1 else L, generator=g).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(DEV) if C2 else None
    gidx = torch.randint(-1, L1 + 1, (B, L), generator=g, dtype=torch.int32).to(DEV) if L1 else None
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    wp = ops.pointmlp_pack(w, "h3")
    y_new, y_old = torch.empty(B, Cout, L, device=DEV), torch.empty(B, Cout, L, device=DEV)
    f_new = lambda: ops.pointmlp(x1, wp, scale, shift, True, Cout, x2=x2, out=y_new, gidx=gidx)   # noqa: E731
    f_old = lambda: ops.pointmlp(x1, wp, scale, shift, True, Cout, x2=x2, out=y_old, gidx=gidx)   # noqa: E731
    iters = 20 if L < 5000 else 5
    os.environ["SONET_POINTMLP_H3R"] = "1"
    t_new = run(f_new, iters)
    os.environ["SONET_POINTMLP_H3R"] = "0"
    t_old = run(f_old, iters)
    os.environ.pop("SONET_POINTMLP_H3R")
    same = torch.equal(y_new, y_old)
    xin = torch.cat([x1 if gidx is None else torch.zeros(B, C1, L, device=DEV), x2], 1) if x2 is not None else x1
    fl = 2.0 * (C1 + C2) * Cout * B * L * 3
    print("%4dx%-4d L=%-5d B=%-2d %s: new %8.4f ms (%5.1f%% of 2.5 PF f16 x3)  old %8.4f ms   bit-identical %s" % (
        C1 + C2, Cout, L, B, "gather" if L1 else "      ", t_new, 100 * fl / t_new / 1e9 / 2.5e3, t_old, same), flush=True)
    if not same:
        d = (y_new - y_old).abs()
        print("    max |diff| %.3g, %d of %d differ, nan %d" % (float(d.max()), int((y_new != y_old).sum()), y_new.numel(), int(torch.isnan(y_new).sum())))
