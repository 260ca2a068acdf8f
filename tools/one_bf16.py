"""tools/one_bf16.py C1 C2 Cout L [iters] -- a few launches of the bf16 layer on one shape at B = 64 (counter runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402

from sonet_hip import ops  # noqa: E402

C1, C2, Cout, L = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
B, DEV = 64, "cuda:0"
g = torch.Generator().manual_seed(0)
x1 = torch.randn(B, C1, L, generator=g).to(DEV).bfloat16()
x2 = torch.randn(B, C2, L, generator=g).to(DEV).bfloat16() if C2 else None
W = (torch.randn(Cout, C1 + C2, generator=g) * (2.0 / (C1 + C2)) ** 0.5).to(DEV)
sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
wp = ops.pointmlp_pack(W, "bf16")
y = torch.empty(B, Cout, L, dtype=torch.bfloat16, device=DEV)
for _ in range(iters):
    ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y)
torch.cuda.synchronize()
