"""tools/fuzz_bf16r.py [cases] [seed] -- random shapes through the streaming bf16 layer kernel (pointmlp_bf16r_kernel: hand-counted waits,
persistent waves) against the staged kernel (the same layer on <= 16 clouds at a time): bit-identical outputs, statistics against f64.
Shapes: K = C1 + C2 a multiple of 64 (C1 a multiple of 16), Cout a multiple of 64, L even with >= 8192 column groups in the batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    K = int(rng.choice([64, 128, 192, 256, 320, 384]))
    C2 = int(rng.choice([0, 0, 64, 128, 256]))
    if C2 >= K:
        C2 = 0
    C1 = K - C2
    Cout = int(rng.choice([64, 128, 192, 256, 384, 512]))
    L = int(rng.integers(1100, 6000)) * 2
    gpc = (L + 63) // 64
    B = max(17, (8192 + gpc - 1) // gpc + int(rng.integers(0, 3)))
    if B * (K + Cout) * L * 2 > 6e9:
        continue
    relu, stats = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(case)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    W = (torch.randn(Cout, K, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    wp = ops.pointmlp_pack(W, "bf16")
    if stats:
        y, m1, v1 = ops.pointmlp_stats(x1, wp, sc, sh, relu, Cout, x2=x2)
    else:
        y = ops.pointmlp(x1, wp, sc, sh, relu, Cout, x2=x2)
    ref = torch.cat([ops.pointmlp(x1[b0:b0 + 16].contiguous(), wp, sc, sh, relu, Cout, x2=x2[b0:b0 + 16].contiguous() if C2 else None)
                     for b0 in range(0, B, 16)], dim=0)
    ok = torch.equal(y, ref)
    msg = ""
    if stats:
        yd = y.double()
        mref, vref = yd.mean(dim=(0, 2)), yd.var(dim=(0, 2), unbiased=False)
        scl = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
        em, ev = float(((m1.double() - mref).abs() / scl).max()), float(((v1.double() - vref).abs() / scl ** 2).max())
        ok = ok and em < 1e-6 and ev < 2e-6
        msg = " stats err %.2g %.2g" % (em, ev)
    bad += not ok
    print("case %2d  B=%3d %3d|%3d -> %3d  L=%5d relu=%d stats=%d : %s%s" % (case, B, C1, C2, Cout, L, relu, stats, "ok" if ok else "MISMATCH %d" % int((y != ref).sum()), msg), flush=True)
    del x1, x2, y, ref
    torch.cuda.empty_cache()
print("fuzz_bf16r: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
