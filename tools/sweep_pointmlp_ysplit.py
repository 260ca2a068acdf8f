"""tools/sweep_pointmlp_ysplit.py -- node-level layer shapes of the classifier forward (B = 64) under the launch knobs of
the layer-wise h3 kernel: SONET_POINTMLP_MT (cout tiles per wave) x SONET_POINTMLP_YSPLIT (output-channel slabs per column
group).  Results are bit-identical across knobs (same per-tile arithmetic); only the work partition changes.
Run on the GPU box:  python tools/sweep_pointmlp_ysplit.py > gpurun_out/sweep_ysplit.log"""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [(64, 387, 512, 576), (64, 512, 512, 576), (64, 515, 768, 64), (64, 768, 1024, 64)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, Cin, Cout, L in SHAPES:
    x = torch.randn(B, Cin, L, device=DEV)
    W = torch.randn(Cout, Cin, device=DEV) * (2.0 / Cin) ** 0.5
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    wp = ops.pointmlp_pack(W, "h3")
    for k in ("SONET_POINTMLP_MT", "SONET_POINTMLP_YSPLIT"):
        os.environ.pop(k, None)
    ref = ops.pointmlp(x, wp, sc, sh, True, Cout).clone()
    base = timed(lambda: ops.pointmlp(x, wp, sc, sh, True, Cout))
    print("%4dx%-4d L=%-4d default            : %7.1f us %6.1f TF-eq" % (Cin, Cout, L, base * 1e3, 2.0 * Cin * Cout * B * L / base / 1e9))
    for mt in ("6", "4", "2"):
        for ys in ("1", "2", "4", "8", "16"):
            os.environ["SONET_POINTMLP_MT"], os.environ["SONET_POINTMLP_YSPLIT"] = mt, ys
            y = ops.pointmlp(x, wp, sc, sh, True, Cout)
            same = torch.equal(y, ref)
            ms = timed(lambda: ops.pointmlp(x, wp, sc, sh, True, Cout))
            print("%4dx%-4d L=%-4d MT=%s YSPLIT=%-2s %s: %7.1f us %6.1f TF-eq  (%+.0f %%)" % (
                Cin, Cout, L, mt, ys, "   " if same else "DIFF", ms * 1e3, 2.0 * Cin * Cout * B * L / ms / 1e9, (ms / base - 1) * 100))
    for k in ("SONET_POINTMLP_MT", "SONET_POINTMLP_YSPLIT"):
        os.environ.pop(k, None)
