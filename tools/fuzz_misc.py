"""tools/fuzz_misc.py [n] [seed] -- random shapes: bf16 layer kernel (+ statistics epilogue) vs float64 on the same bf16 operands;
sonet_wgrad_x3_f32 vs float64; linear_act vs float64; node_gather_lead vs float64."""
import os
import random
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    g = torch.Generator().manual_seed(1000 + it)
    why = []
    # bf16 layer
    B = rng.choice([1, 2, 3, 8]); C1 = rng.choice([6, 16, 64, 100, 128, 320, 512]); C2 = rng.choice([0, 3, 64]) if C1 % 16 == 0 else 0
    Cout = 32 * rng.choice([1, 2, 4, 6, 8, 12, 16, 24]); L = rng.choice([1, 2, 33, 64, 100, 576, 1000, 1029])
    w = (torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(DEV)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    wp = ops.pointmlp_pack(w, "bf16")
    y = ops.pointmlp(x1, wp, scale, shift, True, Cout, x2=x2)
    ys, m, v = ops.pointmlp_stats(x1, wp, scale, shift, True, Cout, x2=x2)
    xin = torch.cat([x1, x2], 1).double() if x2 is not None else x1.double()
    wb = w.to(torch.bfloat16).double()
    ref = torch.relu(torch.einsum("oc,bcl->bol", wb, xin) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1))
    err = (y.double() - ref).abs()
    if not bool((err <= 2.0 ** -7 * ref.abs() + 2.0 ** -9 * max(1e-3, float(ref.pow(2).mean().sqrt()))).all()):
        why.append("bf16 layer worst %.3g" % float(err.max()))
    yd = y.double()
    mref, vref = yd.mean(dim=(0, 2)), yd.var(dim=(0, 2), unbiased=False)
    sc = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
    if not (torch.equal(y, ys) and float(((m.double() - mref).abs() / sc).max()) < 2e-6 and float(((v.double() - vref).abs() / sc ** 2).max()) < 4e-6):
        why.append("bf16 stats")
    # wgrad
    Bw = rng.choice([1, 2, 4]); Co = rng.choice([64, 96, 128, 256, 384, 512]); Ci = rng.choice([3, 6, 33, 64, 128, 320, 387]); Lw = rng.choice([1, 31, 64, 100, 1000, 4099])
    gg = (torch.randn(Bw, Co, Lw, generator=g) * 1e-3).to(DEV)
    xx = torch.randn(Bw, Ci, Lw, generator=g).to(DEV)
    dw = ops.wgrad_x3(gg, xx)
    refw = torch.einsum("bol,bcl->oc", gg.double(), xx.double())
    scw = float(torch.einsum("bol,bcl->oc", gg.double().abs(), xx.double().abs()).max())
    if not float((dw.double() - refw).abs().max()) <= 2e-6 * scw:
        why.append("wgrad err %.3g of %.3g" % (float((dw.double() - refw).abs().max()), scw))
    # linear_act
    Bl = rng.choice([1, 3, 16, 64, 70]); Ki = rng.choice([5, 33, 256, 512, 1024, 1500, 4096]); Ko = rng.choice([1, 5, 40, 256, 512])
    xl, Wl = torch.randn(Bl, Ki, generator=g).to(DEV), (torch.randn(Ko, Ki, generator=g) * 0.05).to(DEV)
    scl, shl = (torch.rand(Ko, generator=g) + 0.5).to(DEV), torch.randn(Ko, generator=g).to(DEV)
    refl = torch.relu((xl.double() @ Wl.double().t()) * scl.double() + shl.double())
    gl = ops.linear_act(xl, Wl, scl, shl, True)
    if not float((gl.double() - refl).abs().max()) <= 4e-6 * max(1.0, float(refl.abs().max())):
        why.append("linear_act")
    if why:
        bad += 1
        print("MISMATCH it=%d (B=%d C1=%d C2=%d Cout=%d L=%d | wgrad %dx%d L=%d | fc %dx%d B=%d): %s" % (it, B, C1, C2, Cout, L, Co, Ci, Lw, Ki, Ko, Bl, "; ".join(why)), flush=True)
print("fuzz_misc: %d rounds, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
