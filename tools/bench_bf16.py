"""tools/bench_bf16.py -- stand-alone timing of the bf16 layer-wise kernel on the first PointNet's layer shapes (B = 64,
L = 15000) and the node-level shapes, per tile variant: algorithmic HBM bytes / time and flops / time."""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)

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


def run_fused(iters=20):
    from models import layers as Lm
    from sonet_hip import synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    x = torch.randn(B, 6, 15000, device=DEV)
    with ops.precision("bf16"), torch.no_grad():
        for _ in range(3):
            pr(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            pr(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * (6 * 64 + 64 * 128 + 128 * 256 + 320 * 384) * B * 15000
    print("fused bf16 first PointNet (store variant) B=%d L=15000: %.4f ms  %.1f TF (%.3f of 2500)  output %.0f GB/s"
          % (B, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, B * 384 * 15000 * 2 / ms / 1e6), flush=True)


def run_pool(iters=20):
    from models import layers as Lm
    from sonet_hip import synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    inp = synth.make_inputs(B, 5000, seed=1, device=DEV)
    a = ops.som_assign(inp["pc"], inp["node"], 3)
    sg = ops.som_sort_group(inp["pc"], inp["sn"], a)
    with ops.precision("bf16"), torch.no_grad():
        wstream, affine = pr._fused_state()
        for _ in range(3):
            ops.pointresnet_bf16_pool(sg, wstream, affine, 64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.pointresnet_bf16_pool(sg, wstream, affine, 64)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * (6 * 64 + 64 * 128 + 128 * 256 + 320 * 384) * B * 15000
    print("fused bf16 first PointNet + per-node max-pool B=%d L=15000: %.4f ms  %.1f TF (%.3f of 2500)" % (B, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500), flush=True)


run_pool()
for a in os.environ.get("POOL_ABL", "").split():
    os.environ["SONET_BF16_FUSED_ABLATE"] = a
    print("  (ablate %s:)" % a, end=" ")
    run_pool()
os.environ.pop("SONET_BF16_FUSED_ABLATE", None)
run_fused()
os.environ["SONET_BF16_FUSED_ABLATE"] = "1"
print("  (no stores:)", end=" ")
run_fused()
os.environ.pop("SONET_BF16_FUSED_ABLATE")
if os.environ.get("LAYERS", "1") != "1":
    sys.exit(0)
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
