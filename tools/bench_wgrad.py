"""tools/bench_wgrad.py -- sonet_wgrad_x3_f32 against torch.bmm(g, x^T).sum(0) (hipBLASLt f32) at the training step's dense
weight-gradient shapes (B = 64); run under tools/prof_wgrad.sh for kernel times."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [(64, 64, 6, 15000), (64, 128, 64, 15000), (64, 256, 128, 15000), (64, 512, 387, 576), (64, 512, 512, 576),
          (64, 768, 515, 64), (64, 1024, 768, 64)]
gen = torch.Generator().manual_seed(0)
for B, Cout, Cin, L in (SHAPES if len(sys.argv) < 2 else [SHAPES[int(sys.argv[1])]]):
    g = (torch.randn(B, Cout, L, generator=gen) * 1e-3).to(DEV)
    x = torch.randn(B, Cin, L, generator=gen).to(DEV)
    for _ in range(6):
        a = ops.wgrad_x3(g, x)
    for _ in range(6):
        b = torch.bmm(g, x.transpose(1, 2)).sum(0)
    torch.cuda.synchronize()
    ref = torch.einsum("bol,bcl->oc", g[:4].double(), x[:4].double()) if L > 1000 else torch.einsum("bol,bcl->oc", g.double(), x.double())
    if L > 1000:
        a4, b4 = ops.wgrad_x3(g[:4].contiguous(), x[:4].contiguous()), torch.bmm(g[:4], x[:4].transpose(1, 2)).sum(0)
    else:
        a4, b4 = a, b
    sc = float(ref.abs().max())
    print("%dx%d L=%d: kernel err %.2e  bmm err %.2e  (of max |dw|)" % (Cout, Cin, L, float((a4.double() - ref).abs().max()) / sc,
                                                                      float((b4.double() - ref).abs().max()) / sc), flush=True)
