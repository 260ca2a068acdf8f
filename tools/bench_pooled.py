"""tools/bench_pooled.py -- the sparse backward of the pooled last layer (pooled_dgrad f32 / bf16 output, pooled_wgrad) at the
benchmark shape, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
B, C, M, C1, C2, L = 64, 384, 64, 64, 256, 15000
g = torch.Generator().manual_seed(0)
gp = (torch.randn(B, C, M, generator=g) * 1e-3).to(DEV)
pos = torch.randint(0, L, (B, C, M), generator=g, dtype=torch.int32).to(DEV)
w = torch.randn(C, C1 + C2, generator=g).to(DEV)
x1 = torch.randn(B, C1, L, generator=g).to(DEV)
x2 = torch.randn(B, C2, L, generator=g).to(DEV)
for _ in range(5):
    a = ops.pooled_dgrad(gp, pos, w, C1, C2, L)
for _ in range(5):
    b = ops.pooled_dgrad(gp, pos, w, C1, C2, L, out_dtype=torch.bfloat16)
gt, pt = gp.transpose(1, 2).contiguous(), pos.transpose(1, 2).contiguous()
for _ in range(5):
    ops.pooled_wgrad(gt, pt, x1)
    ops.pooled_wgrad(gt, pt, x2)
torch.cuda.synchronize()
print("bf16 vs f32 output: max rel diff %.3g" % float(((b[1].float() - a[1]).abs() / a[1].abs().max()).max()))
