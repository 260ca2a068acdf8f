"""tools/bench_bf16_xaff.py -- what normalise-on-load costs the bf16 consumers (B = 64 x 15000 columns): the layer kernels of the first PointNet
fed the normalised tensor against the same kernels normalising RAW inputs in their operand loads, next to the normalise pass they make unnecessary."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
B, L, M = 64, 15000, 64


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


def raw(C, seed):
    g = torch.Generator().manual_seed(seed)
    return ((torch.randn(B, C, L, generator=g) * 1.5).to(torch.bfloat16).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV),
            (torch.randn(C, generator=g) * 0.4).to(DEV))


with ops.precision("bf16"):
    for C1, C2, Cout in [(64, 0, 128), (128, 0, 256)]:
        r1, s1, h1 = raw(C1, C1)
        W = (torch.randn(Cout, C1) * C1 ** -0.5).to(DEV)
        bias = torch.zeros(Cout, device=DEV)
        wp, one = ops.pointmlp_pack(W, "bf16"), ops.const_vec(Cout, 1.0, DEV)
        y1 = ops.channel_affine_act(r1, s1, h1, True)
        a = t(lambda: ops.channel_affine_act(r1, s1, h1, True))
        b = t(lambda: ops.pointmlp_stats(y1, wp, one, bias, False, Cout))
        c = t(lambda: ops.pointmlp_stats(r1, wp, one, bias, False, Cout, xaff=(s1, h1, True)))
        print("%3d -> %3d + stats: normalise pass %6.1f us | layer %6.1f us | layer normalising on load %6.1f us" % (C1, Cout, a, b, c))
    r1, s1, h1 = raw(64, 1)
    r2, s2, h2 = raw(256, 2)
    W = (torch.randn(384, 320) * 320 ** -0.5).to(DEV)
    bias = torch.zeros(384, device=DEV)
    wp, one = ops.pointmlp_pack(W, "bf16"), ops.const_vec(384, 1.0, DEV)
    ids = torch.randint(0, M, (B, L), dtype=torch.int32).to(DEV)
    rm = torch.ones(B, M, dtype=torch.int32, device=DEV)
    y1, y2 = ops.channel_affine_act(r1, s1, h1, True), ops.channel_affine_act(r2, s2, h2, True)
    a = t(lambda: (ops.channel_affine_act(r1, s1, h1, True), ops.channel_affine_act(r2, s2, h2, True)))
    b = t(lambda: ops.pointmlp_bf16_pool(y1, wp, one, bias, False, 384, ids, M, rm, x2=y2))
    c = t(lambda: ops.pointmlp_bf16_pool(r1, wp, one, bias, False, 384, ids, M, rm, x2=r2, xaff=(s1, h1, True, s2, h2, True)))
    print("64|256 -> 384 + pool: normalise passes %6.1f us | layer %6.1f us | layer normalising on load %6.1f us" % (a, b, c))
    for Cout, Cin in [(128, 64), (256, 128)]:
        rw, sc, sh = raw(Cin, Cin + 7)
        g = (torch.randn(B, Cout, L) * 1e-3).to(torch.bfloat16).to(DEV)
        y = ops.channel_affine_act(rw, sc, sh, True)
        b = t(lambda: ops.wgrad_bf16(g, y))
        c = t(lambda: ops.wgrad_bf16(g, rw, xaff=(sc, sh, True)))
        print("wgrad %3d x %3d: %6.1f us | normalising on load %6.1f us" % (Cout, Cin, b, c))
    g_t = torch.randn(B, M, 384).to(DEV)
    pos_t = torch.randint(0, L, (B, M, 384), dtype=torch.int32).to(DEV)
    for Ci in (64, 256):
        rw, sc, sh = raw(Ci, Ci + 9)
        y = ops.channel_affine_act(rw, sc, sh, True)
        b = t(lambda: ops.pooled_wgrad(g_t, pos_t, y))
        c = t(lambda: ops.pooled_wgrad(g_t, pos_t, rw, (sc, sh, True)))
        print("pooled wgrad 384 x %3d: %6.1f us | normalising on load %6.1f us" % (Ci, b, c))
