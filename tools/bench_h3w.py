"""tools/bench_h3w.py -- the fp16-split layer kernel with two column tiles per wave (pointmlp_h3r_kernel<2>) against one column
tile (SONET_POINTMLP_NC=1: whatever the dispatcher picked before, second or first generation) on the point-level shapes: outputs
must be bit-identical; time per launch and fraction of the nominal 2.5 PFLOP/s / 3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs live in the variants build)

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # (B, C1, C2, Cout, L)
    (64, 1024, 0, 512, 3072), (64, 512, 0, 256, 3072), (64, 393, 0, 1024, 3072), (64, 256, 0, 128, 3072),
    (64, 320, 0, 384, 15000), (64, 256, 64, 384, 15000), (64, 128, 0, 256, 15000), (64, 64, 0, 128, 15000), (64, 6, 0, 64, 15000),
    (64, 512, 0, 512, 576), (16, 320, 0, 384, 5000), (8, 1347, 0, 512, 1024), (3, 40, 0, 128, 78),
]


def run(fn, iters):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B, C1, C2, Cout, L in SHAPES:
    g = torch.Generator().manual_seed(C1 + Cout + L)
    w = (torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(DEV)
    x1 = torch.randn(B, C1, L, generator=g).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(DEV) if C2 else None
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    wp = ops.pointmlp_pack(w, "h3")
    ys = {"1": torch.empty(B, Cout, L, device=DEV), "2": torch.empty(B, Cout, L, device=DEV)}
    ts = {"1": [], "2": []}
    iters = 20 if B * L < 300000 else 8
    for nc in ("2", "1", "2", "1", "2", "1"):                 # alternate: the clock a launch sees depends on what ran before it
        os.environ["SONET_POINTMLP_NC"] = nc
        ts[nc].append(run(lambda: ops.pointmlp(x1, wp, scale, shift, True, Cout, x2=x2, out=ys[nc]), iters))
    os.environ.pop("SONET_POINTMLP_NC")
    fl = 2.0 * (C1 + C2) * Cout * B * L
    t2, t1 = min(ts["2"]), min(ts["1"])
    print("%4dx%-4d L=%-5d B=%-2d: two tiles %8.4f ms (%4.1f%% of 833 TF)   one tile %8.4f ms (%4.1f%%)   ratio %.3f   bit-identical %s" % (
        C1 + C2, Cout, L, B, t2, 100 * fl / t2 / 1e9 / 833.3, t1, 100 * fl / t1 / 1e9 / 833.3, t2 / t1, torch.equal(ys["2"], ys["1"])), flush=True)
