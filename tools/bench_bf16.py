"""tools/bench_bf16.py -- stand-alone timing of the bf16 layer-wise kernel on the first PointNet's layer shapes (B = 64,
L = 15000) and the node-level shapes, per tile variant: algorithmic HBM bytes / time and flops / time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))


def run(C1, C2, Cout, L, iters=20):
    x1 = torch.randn(B, C1, L, device=DEV).to(torch.bfloat16)
    x2 = torch.randn(B, C2, L, device=DEV).to(torch.bfloat16) if C2 else None
    wp = ops.pointmlp_pack(torch.randn(Cout, C1 + C2, device=DEV) * 0.08, "bf16")
    sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
    y = torch.empty(B, Cout, L, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    byt = B * L * 2.0 * (C1 + C2 + Cout)
    fl = 2.0 * B * L * (C1 + C2) * Cout
    return ms, byt / ms / 1e6, fl / ms / 1e9


shapes = [(6, 0, 64, 15000), (64, 0, 128, 15000), (128, 0, 256, 15000), (64, 256, 384, 15000), (384, 0, 512, 576), (512, 0, 512, 576),
          (512, 0, 768, 64), (768, 0, 1024, 64)]
for C1, C2, Cout, L in shapes:
    os.environ.pop("SONET_BF16_MT", None), os.environ.pop("SONET_BF16_S", None)
    for xr in ("1", "0"):
        os.environ["SONET_BF16_XREG"] = xr
        ms, gbs, tf = run(C1, C2, Cout, L)
        print("%4d+%-3d -> %-4d L=%-5d default XREG=%s  %.4f ms  %7.0f GB/s (%.2f of 8000)  %6.1f TF (%.3f of 2500)"
              % (C1, C2, Cout, L, xr, ms, gbs, gbs / 8000, tf, tf / 2500), flush=True)
    os.environ.pop("SONET_BF16_XREG")
    if os.environ.get("SWEEP", "0") != "1":
        continue
    CT = Cout // 32
    for mt in ("12", "6", "4", "2"):
        if CT % int(mt):
            continue
        for s in ("1", "2"):
            if mt == "12" and s == "2":
                continue
            os.environ["SONET_BF16_MT"], os.environ["SONET_BF16_S"] = mt, s
            ms, gbs, tf = run(C1, C2, Cout, L)
            print("%4d+%-3d -> %-4d L=%-5d MT=%-2s S=%s  %.4f ms  %7.0f GB/s (%.2f of 8000)  %6.1f TF (%.3f of 2500)"
                  % (C1, C2, Cout, L, mt, s, ms, gbs, gbs / 8000, tf, tf / 2500), flush=True)
os.environ.pop("SONET_BF16_MT", None), os.environ.pop("SONET_BF16_S", None)
