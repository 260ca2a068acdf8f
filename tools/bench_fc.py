"""tools/bench_fc.py -- sonet_linear_act_f32 at the classifier-head shapes (run under rocprofv3 --kernel-trace --stats for kernel times)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
for B, Cin, Cout in [(64, 1024, 512), (64, 512, 256), (64, 256, 40), (64, 1024, 1024), (8, 1024, 512)]:
    x, W = torch.randn(B, Cin, generator=g).to(DEV), (torch.randn(Cout, Cin, generator=g) * 0.05).to(DEV)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    ref = torch.relu((x.double() @ W.double().t()) * sc.double() + sh.double())
    for _ in range(20):
        got = ops.linear_act(x, W, sc, sh, True)
    torch.cuda.synchronize()
    print("%dx%d B=%d max err %.3g" % (Cin, Cout, B, float((got.double() - ref).abs().max())), flush=True)
