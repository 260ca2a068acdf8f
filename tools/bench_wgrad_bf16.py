"""tools/bench_wgrad_bf16.py -- sonet_wgrad_bf16 against torch.bmm(g, x^T, out_dtype=f32).sum(0) (hipBLASLt) on the weight-gradient
shapes of the bf16 training step (B = 64)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [(64, 6, 15000), (128, 64, 15000), (256, 128, 15000), (512, 387, 576), (512, 512, 576), (768, 515, 64), (1024, 768, 64)]


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for Cout, Cin, L in SHAPES:
    g = torch.Generator().manual_seed(1)
    gg = (torch.randn(64, Cout, L, generator=g) * 1e-3).to(torch.bfloat16).to(DEV)
    xx = torch.randn(64, Cin, L, generator=g).to(torch.bfloat16).to(DEV)
    a = t(lambda: ops.wgrad_bf16(gg, xx))
    b = t(lambda: torch.bmm(gg, xx.transpose(1, 2), out_dtype=torch.float32).sum(0))
    gb = (Cout + Cin) * 64 * L * 2 / 1e9
    print("%4d x %4d  L=%5d : wgrad_bf16 %7.1f us (%.2f TB/s of operand bytes)   hipBLASLt bmm + sum %7.1f us" % (Cout, Cin, L, a, gb / a * 1e3, b))
