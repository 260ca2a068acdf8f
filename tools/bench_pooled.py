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
# one position per node stratum: the positions of a (cloud, channel) row are distinct, as the arg-max positions of different nodes are
pos = (torch.arange(M).view(1, 1, M) * (L // M) + torch.randint(0, L // M, (B, C, M), generator=g)).to(torch.int32).to(DEV)
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


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


x1b, x2b = x1.bfloat16(), x2.bfloat16()
print("pooled_dgrad f32 out  %.4f ms" % timeit(lambda: ops.pooled_dgrad(gp, pos, w, C1, C2, L)))
print("pooled_dgrad bf16 out %.4f ms" % timeit(lambda: ops.pooled_dgrad(gp, pos, w, C1, C2, L, out_dtype=torch.bfloat16)))
wtp = ops.pointmlp_pack(w.t().contiguous(), "bf16")
print("pooled_dgrad mfma bf16 %.4f ms" % timeit(lambda: ops.pooled_dgrad(gp, pos, w, C1, C2, L, out_dtype=torch.bfloat16, wt_pack=wtp)))
m = ops.pooled_dgrad(gp, pos, w, C1, C2, L, out_dtype=torch.bfloat16, wt_pack=wtp)
for k in (0, 1):
    d = (m[k].float() - a[k]).abs().max() / a[k].abs().max()
    print("  mfma vs f32 scalar, part %d: max |diff| / max |ref| = %.3g; equal run to run: %s" % (k, float(d), torch.equal(m[k], ops.pooled_dgrad(gp, pos, w, C1, C2, L, out_dtype=torch.bfloat16, wt_pack=wtp)[k])))
print("pooled_wgrad f32 x1 / x2   %.4f / %.4f ms" % (timeit(lambda: ops.pooled_wgrad(gt, pt, x1)), timeit(lambda: ops.pooled_wgrad(gt, pt, x2))))
print("pooled_wgrad bf16 x1 / x2  %.4f / %.4f ms" % (timeit(lambda: ops.pooled_wgrad(gt, pt, x1b)), timeit(lambda: ops.pooled_wgrad(gt, pt, x2b))))
print("bf16 vs f32 output: max rel diff %.3g" % float(((b[1].float() - a[1]).abs() / a[1].abs().max()).max()))
