"""tools/fuzz_train_kernels.py -- random shapes through the training kernels of this round that the parametrized tests pin at a few
shapes only, each against float64 on the same rounded operands:

  wgrad   sonet_wgrad_bf16 (both generations: the dispatch picks by shape)       sum_b g[b] x[b]^T
  pdgrad  sonet_pooled_dgrad_mfma_bf16 (both tile widths, SONET_PM_CH needs the variants build) and the scalar f32 kernel
  adam    sonet_adam_multi_f32 against torch.optim.Adam on ragged tensor lists (sizes around the chunk length)

  python tools/fuzz_train_kernels.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))

import torch  # noqa: E402

from sonet_hip import _lib, ops  # noqa: E402
from sonet_hip.optim import FusedAdam  # noqa: E402

DEV = "cuda:0"


def rint(gen, lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


def fuzz_wgrad(gen, i):
    big = i % 2 == 0                                                    # the streaming generation needs B x ceil(L / 64) >= 2048 and L % 8 == 0
    Cout, Cin = rint(gen, 1, 400), rint(gen, 1, 400)
    if big:
        L = 8 * rint(gen, 40, 1900)
        B = max(1, (2048 * 64 + L - 1) // L + rint(gen, 0, 3))
    else:
        L, B = rint(gen, 1, 700), rint(gen, 1, 9)
    g = (torch.randn(B, Cout, L, generator=gen) * 1e-2).to(torch.bfloat16).to(DEV)
    x = torch.randn(B, Cin, L, generator=gen).to(torch.bfloat16).to(DEV)
    got = ops.wgrad_bf16(g, x)
    ref = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
    for b0 in range(0, B, 8):
        ref += torch.bmm(g[b0:b0 + 8].double(), x[b0:b0 + 8].double().transpose(1, 2)).sum(0)
    err = float((got.double() - ref).abs().max())
    scale = float(ref.abs().max()) + 1e-30
    rms = float(((got.double() - ref) ** 2).mean().sqrt() / ((ref ** 2).mean().sqrt() + 1e-30))
    same = torch.equal(got, ops.wgrad_bf16(g, x))
    ok = rms <= 2e-5 and err <= 2e-4 * scale and same
    return ok, "wgrad  B=%d Cout=%d Cin=%d L=%d: rms %.2e max/scale %.2e deterministic %s" % (B, Cout, Cin, L, rms, err / scale, same)


def fuzz_pdgrad(gen, i):
    C = 16 * rint(gen, 1, 24)
    ct = 2 * rint(gen, 1, 6)
    Cin = rint(gen, 32 * (ct - 1) + 1, 32 * ct)
    C1 = rint(gen, 1, Cin) if i % 3 else Cin
    C2 = Cin - C1
    M = rint(gen, 1, 80)
    big = i % 4 == 0                                                    # every fourth case: the 128-column tiles (B x ceil(L / 128) >= 2048)
    L = 2 * rint(gen, max(M, 500 if big else 33), 3500)
    B = max(1, 2048 * 128 // L + 1) if big else rint(gen, 1, 5)
    g = torch.randn(B, C, M, generator=gen)
    pos = (torch.arange(M).view(1, 1, M) * (L // M) + torch.randint(0, L // M, (B, C, M), generator=gen)).to(torch.int32)
    pos[torch.rand(B, C, M, generator=gen) < 0.05] = -1
    W = torch.randn(C, Cin, generator=gen) * 0.1
    wt = W.t().contiguous()
    if wt.shape[0] % 32:
        wt = torch.cat((wt, wt.new_zeros(32 - wt.shape[0] % 32, C)), dim=0)
    wtp = ops.pointmlp_pack(wt.to(DEV), "bf16")
    if not ops.pooled_dgrad_mfma_ok(C, C1, C2, L):
        return False, "pdgrad shape generator left the kernel's domain: C=%d C1=%d C2=%d L=%d" % (C, C1, C2, L)
    gd, pd, Wd = g.to(DEV), pos.to(DEV), W.to(DEV)

    def cat(t):
        return torch.cat([t[0]] + ([t[1]] if C2 else []), dim=1)

    got = cat(ops.pooled_dgrad(gd, pd, Wd, C1, C2, L, out_dtype=torch.bfloat16, wt_pack=wtp))
    again = cat(ops.pooled_dgrad(gd, pd, Wd, C1, C2, L, out_dtype=torch.bfloat16, wt_pack=wtp))
    got32 = cat(ops.pooled_dgrad(gd, pd, Wd, C1, C2, L, out_dtype=torch.float32))
    ok, worst, worst32 = True, 0.0, 0.0
    for b0 in range(0, B, 8):
        pp = pd[b0:b0 + 8]
        idx = torch.where(pp < 0, L, pp).long()
        Gb = torch.zeros(pp.shape[0], C, L + 1, dtype=torch.float64, device=DEV).scatter_add_(2, idx, gd[b0:b0 + 8].to(torch.bfloat16).double())[:, :, :L]
        want = torch.matmul(Wd.to(torch.bfloat16).double().t().unsqueeze(0), Gb)
        err = (got[b0:b0 + 8].double() - want).abs()
        ok = ok and bool((err <= want.abs() * 2.0 ** -8 + 1e-6 * float(want.abs().max())).all())
        worst = max(worst, float(err.max()) / (float(want.abs().max()) + 1e-30))
        Gf = torch.zeros(pp.shape[0], C, L + 1, dtype=torch.float64, device=DEV).scatter_add_(2, idx, gd[b0:b0 + 8].double())[:, :, :L]
        want32 = torch.matmul(Wd.double().t().unsqueeze(0), Gf)
        e32 = float((got32[b0:b0 + 8].double() - want32).abs().max()) / (float(want32.abs().max()) + 1e-30)
        worst32 = max(worst32, e32)
        ok = ok and e32 <= 1e-5
    same = torch.equal(got, again)
    return ok and same, "pdgrad B=%d C=%d M=%d C1=%d C2=%d L=%d: bf16 max/scale %.2e  f32 kernel max/scale %.2e  deterministic %s" % (
        B, C, M, C1, C2, L, worst, worst32, same)


def fuzz_adam(gen, i):
    chunk = int(_lib.load().sonet_adam_chunk())
    T = rint(gen, 1, 24)
    sizes = [rint(gen, 1, 3 * chunk) if rint(gen, 0, 2) else chunk * rint(gen, 1, 3) + rint(gen, -1, 1) for _ in range(T)]
    lr, b1, b2, eps = 10.0 ** -rint(gen, 2, 4), [0.9, 0.7, 0.0][i % 3], [0.999, 0.99, 0.9][i % 3], [1e-8, 1e-6][i % 2]
    p0 = [torch.randn(n, generator=gen) for n in sizes]
    pa = [torch.nn.Parameter(p.clone().to(DEV)) for p in p0]
    pb = [torch.nn.Parameter(p.clone().to(DEV)) for p in p0]
    oa, ob = FusedAdam(pa, lr=lr, betas=(b1, b2), eps=eps), torch.optim.Adam(pb, lr=lr, betas=(b1, b2), eps=eps)
    worst = 0.0
    for s in range(4):
        for t, (a, b) in enumerate(zip(pa, pb)):
            if (t + s) % 5 == 4:                                         # a parameter without a gradient this step
                a.grad = b.grad = None
                continue
            gr = (torch.randn(a.numel(), generator=gen) * 10.0 ** rint(gen, -4, 1)).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            worst = max(worst, float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30)))
    return worst <= 2e-6, "adam   T=%d sizes %d..%d lr=%g betas=(%g, %g) eps=%g: max |p - torch| / max |p| after 4 steps %.2e" % (
        T, min(sizes), max(sizes), lr, b1, b2, eps, worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    gen = torch.Generator().manual_seed(args.seed)
    bad = 0
    for name, fn in (("wgrad", fuzz_wgrad), ("pdgrad", fuzz_pdgrad), ("adam", fuzz_adam)):
        if args.only and name not in args.only.split(","):
            continue
        for i in range(args.cases):
            ok, line = fn(gen, i)
            print(("ok   " if ok else "FAIL ") + line, flush=True)
            bad += 0 if ok else 1
            torch.cuda.empty_cache()
    print("%d failing case(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
