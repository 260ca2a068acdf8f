"""tools/bench_bwd_passes.py -- the element-wise passes of the training step (BatchNorm / ReLU backward statistics and apply, forward
statistics, normalise + ReLU) on B x C x 15000 tensors, bf16 and f32, against the HBM roofline (bytes read + written, once)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV, B, L = "cuda:0", 64, 15000


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
    return e0.elapsed_time(e1) / it


for dt in (torch.bfloat16, torch.float32):
    eb = 2 if dt == torch.bfloat16 else 4
    for C in (64, 128, 256):
        g = torch.Generator().manual_seed(C)
        gy = torch.randn(B, C, L, generator=g).to(dt).to(DEV)
        raw = torch.randn(B, C, L, generator=g).to(dt).to(DEV)
        sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
        a, b, c0 = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        n = B * C * L
        rows = [("bwd stats (2 reads)", lambda: ops.pointwise_bwd_stats(gy, raw, sc, sh, True), 2 * n * eb),
                ("bwd apply (2 reads, 1 write)", lambda: ops.pointwise_bwd_apply(gy, raw, sc, sh, True, a, b, c0), 3 * n * eb),
                ("fwd stats (1 read)", lambda: ops.channel_stats(raw), n * eb),
                ("normalise + ReLU (1 read, 1 write)", lambda: ops.channel_affine_act(raw, sc, sh, True), 2 * n * eb)]
        for name, fn, byt in rows:
            ms = t(fn)
            print("%-8s C=%3d  %-36s %7.4f ms  %5.2f TB/s" % (str(dt).split(".")[-1], C, name, ms, byt / ms / 1e9), flush=True)
        del gy, raw
        torch.cuda.empty_cache()
