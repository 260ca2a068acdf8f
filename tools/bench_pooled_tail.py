"""tools/bench_pooled_tail.py -- the sparse input gradient of the pooled layer at the benchmark shape (64 x 384 x 64 entries, 320 input channels,
15000 columns; one position per node stratum) followed by what the training step runs behind it -- the column-0 additions and the
BatchNorm-backward statistics pass of the 256-channel layer -- as launches of their own, against the same work on the store of the launch
(sonet_pooled_dgrad_tail_f32).  Seconds, no model: sized for the last GPU seconds of a round."""
import os
import sys

import _variants  # noqa: F401  (sonet_pooled_dgrad_tail_f32 is a variants-build record since round 6)
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
from sonet_hip import ops  # noqa: E402

dev = "cuda:0"


def check(B, C, M, C1, C2, L, f64):
    """the assertions of tests/test_gpu_segpool.py::test_pooled_dgrad_with_its_tail_equals_dgrad_then_scatter_and_statistics (f64: also the float64
    reference; the big shape compares with the statistics pass only)"""
    gen = torch.Generator().manual_seed(B + L + C2)
    g = (torch.randn(B, C, M, generator=gen) * 1e-2).to(dev)
    pos = torch.randint(0, L, (B, C, M), generator=gen, dtype=torch.int32)
    pos[:, : C // 4, 0] = L - 1
    pos[:, 0, 1] = -1
    pos = pos.to(dev)
    W = (torch.randn(C, C1 + C2, generator=gen) * 0.1).to(dev)
    col0 = (torch.randn(B, C1 + C2, generator=gen) * 1e-2).to(dev)
    p0 = torch.randint(0, L, (B,), generator=gen, dtype=torch.int32)
    p0[0] = L - 1
    p0 = p0.to(dev)
    r1, r2 = ops.pooled_dgrad(g, pos, W, C1, C2, L)
    p0l = p0.long().view(B, 1, 1)
    r1.scatter_add_(2, p0l.expand(B, C1, 1), col0[:, :C1].unsqueeze(2))
    if C2:
        r2.scatter_add_(2, p0l.expand(B, C2, 1), col0[:, C1:].unsqueeze(2))
    a1, a2 = ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0, pos0=p0)
    ok = torch.equal(a1, r1) and (C2 == 0 or torch.equal(a2, r2))
    worst = 0.0
    if C2:
        raw = torch.randn(B, C2, L, device=dev) * 1.5
        sc, sh = torch.rand(C2, device=dev) + 0.5, torch.randn(C2, device=dev) * 0.3
        for relu in (True, False):
            b1, b2, sums = ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0, pos0=p0, below=(raw, sc, sh, relu))
            ok = ok and torch.equal(b1, r1) and torch.equal(b2, r2)
            other = ops.pointwise_bwd_stats(r2, raw, sc, sh, relu, want_sums=True)
            if f64:
                keep = (raw.double() * sc.double().view(1, C2, 1) + sh.double().view(1, C2, 1)) > 0 if relu else torch.ones_like(raw, dtype=torch.bool)
                gm = torch.where(keep, r2.double(), torch.zeros((), dtype=torch.float64, device=dev))
                ref = torch.cat((gm.sum(dim=(0, 2)), (gm * raw.double()).sum(dim=(0, 2))))
                scale = torch.cat((gm.abs().sum(dim=(0, 2)), (gm * raw.double()).abs().sum(dim=(0, 2)))) + 1e-30
                worst = max(worst, float(((sums - ref).abs() / scale).max()))
            else:
                scale = torch.cat((r2.abs().double().sum(dim=(0, 2)), (r2 * raw).abs().double().sum(dim=(0, 2)))) + 1e-30
            worst = max(worst, float(((sums - other).abs() / scale).max()))
            again = ops.pooled_dgrad(g, pos, W, C1, C2, L, col0=col0, pos0=p0, below=(raw, sc, sh, relu))[2]
            ok = ok and torch.equal(again, sums)
    print("check %s: gradients bit-equal and sums reproducible: %s; sums worst deviation / sum of |terms| = %.2e (bound 2e-6)" % ((B, C, M, C1, C2, L), ok, worst), flush=True)
    return ok and worst <= 2e-6


good = all([check(3, 384, 64, 64, 256, 3000, True), check(2, 96, 8, 16, 48, 130, True), check(2, 48, 7, 44, 0, 257, True), check(64, 384, 64, 64, 256, 15000, False)])
print("checks: %s" % ("ok" if good else "FAILED"), flush=True)
B, C, M, C1, C2, L = 64, 384, 64, 64, 256, 15000
g = torch.randn(B, C, M, device=dev) * 1e-3
pos = (torch.arange(M, device=dev).view(1, 1, M) * (L // M) + torch.randint(0, L // M, (B, C, M), device=dev)).to(torch.int32)
w = torch.randn(C, C1 + C2, device=dev) * 0.1
raw2 = torch.randn(B, C2, L, device=dev)
sc2, sh2 = torch.rand(C2, device=dev) + 0.5, torch.randn(C2, device=dev) * 0.3
col0 = torch.randn(B, C1 + C2, device=dev) * 1e-4
p0 = torch.randint(0, L, (B,), device=dev, dtype=torch.int32)
p0l = p0.long().view(B, 1, 1)


def apart():
    a, b2 = ops.pooled_dgrad(g, pos, w, C1, C2, L)
    a.scatter_add_(2, p0l.expand(B, C1, 1), col0[:, :C1].unsqueeze(2))
    b2.scatter_add_(2, p0l.expand(B, C2, 1), col0[:, C1:].unsqueeze(2))
    return ops.pointwise_bwd_stats(b2, raw2, sc2, sh2, True, want_sums=True)


def timeit(fn, iters=8):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rnd in range(2):   # (two rounds: the first warms the clocks)
    t_p = timeit(lambda: ops.pooled_dgrad(g, pos, w, C1, C2, L))
    t_a = timeit(apart)
    t_b = timeit(lambda: ops.pooled_dgrad(g, pos, w, C1, C2, L, col0=col0, pos0=p0, below=(raw2, sc2, sh2, True)))
    t_c = timeit(lambda: ops.pooled_dgrad(g, pos, w, C1, C2, L, col0=col0, pos0=p0))
    print("round %d: pooled_dgrad alone %.4f ms | + 2 scatter_add + statistics pass (256 channels) %.4f ms | on the store of the launch %.4f ms (column 0 only %.4f ms)"
          % (rnd, t_p, t_a, t_b, t_c), flush=True)
