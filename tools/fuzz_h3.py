"""tools/fuzz_h3.py [n] [seed] -- random shapes through every fp16-split layer path: second-generation vs first-generation kernel
bit-identical; statistics / node-add / gather variants vs their two-launch forms; all vs float64 at the f32-class bound."""
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
    B = rng.choice([1, 2, 3, 5, 8])
    C1 = rng.choice([3, 6, 16, 17, 40, 64, 100, 128, 129, 320, 387, 512, 515])
    C2 = rng.choice([0, 0, 3, 9, 64]) if C1 % 16 == 0 else 0
    Cout = 32 * rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24])
    L = rng.choice([1, 7, 31, 32, 33, 64, 100, 129, 576, 1000, 1029, 3072])
    g = torch.Generator().manual_seed(it)
    w = (torch.randn(Cout, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(DEV)
    x1 = torch.randn(B, C1, L, generator=g).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(DEV) if C2 else None
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    relu = rng.random() < 0.5
    wp = ops.pointmlp_pack(w, "h3")
    out = {}
    for gen in ("1", "0"):
        os.environ["SONET_POINTMLP_H3R"] = gen
        out[gen] = ops.pointmlp(x1, wp, scale, shift, relu, Cout, x2=x2)
    os.environ.pop("SONET_POINTMLP_H3R")
    dflt = ops.pointmlp(x1, wp, scale, shift, relu, Cout, x2=x2)
    xin = torch.cat([x1, x2], 1).double() if x2 is not None else x1.double()
    ref = torch.einsum("oc,bcl->bol", w.double(), xin) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1)
    ref = torch.relu(ref) if relu else ref
    tol = 1e-5 * max(1.0, float(ref.abs().max()))
    why = []
    ok = torch.equal(out["1"], out["0"]) and torch.equal(dflt, out["0"]) and float((dflt.double() - ref).abs().max()) <= tol
    if not ok:
        why.append("layer")
    # statistics epilogue
    y, m, v = ops.pointmlp_stats(x1, wp, scale, shift, relu, Cout, x2=x2)
    yd = out["0"].double()                                     # (statistics of the STORED values: the layer's own rounding is not the epilogue's)
    mref, vref = yd.mean(dim=(0, 2)), yd.var(dim=(0, 2), unbiased=False)
    sc = (mref.abs() + vref.sqrt()).clamp_min(1e-3)
    em, ev = float(((m.double() - mref).abs() / sc).max()), float(((v.double() - vref).abs() / sc ** 2).max())
    if not (torch.equal(y, out["0"]) and em < 2e-5 and ev < 4e-5):
        ok = False
        why.append("stats eq=%s em=%.2e ev=%.2e" % (torch.equal(y, out["0"]), em, ev))
    # node-add epilogue
    M = rng.choice([1, 5, 64])
    z = torch.randn(B, Cout, M, generator=g).to(DEV)
    idx = torch.randint(-1, M + 1, (B, L), generator=g, dtype=torch.int32).to(DEV)
    na = ops.pointmlp_nodeadd(x1, wp, scale, shift, relu, Cout, z, idx, x2=x2)
    okm = (idx >= 0) & (idx < M)
    zg = torch.gather(z.double(), 2, idx.clamp(0, M - 1).long().unsqueeze(1).expand(B, Cout, L)) * okm.unsqueeze(1)
    ref2 = (torch.einsum("oc,bcl->bol", w.double(), xin) + zg) * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1)
    ref2 = torch.relu(ref2) if relu else ref2
    e2 = float((na.double() - ref2).abs().max())
    if not e2 <= 1e-5 * max(1.0, float(ref2.abs().max())):
        ok = False
        why.append("nodeadd err %.2e max %.2e" % (e2, float(ref2.abs().max())))
    if not ok:
        bad += 1
        print("MISMATCH B=%d C1=%d C2=%d Cout=%d L=%d relu=%s: %s" % (B, C1, C2, Cout, L, relu, "; ".join(why)), flush=True)
print("fuzz_h3: %d shapes, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
