"""tools/fuzz_bf16_r6.py [cases] [seed] -- random shapes through round 6's bf16 training kernels, each against the launches it replaces, bit for bit:
  * sonet_pointmlp_bf16_bnb (BatchNorm / ReLU backward on the operand load of the input gradient, g_raw on the side, optional accumulating store)
    vs sonet_pointwise_bwd_apply_bf16 + sonet_pointmlp_bf16[_acc] -- any C % 16 == 0 in [32, 512], Cout % 64 == 0, even L, any B;
  * sonet_pointmlp_bf16_stats_xaff / sonet_wgrad_bf16_xaff (normalise-on-load on the streaming kernels: >= 8192 column groups in the batch)
    vs the same launch on sonet_channel_affine_act_bf16's output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def eq(a, b):
    return torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)


with ops.precision("bf16"):
    for case in range(cases):
        g = torch.Generator().manual_seed(1000 + case)
        # ---- bnb
        C = 16 * int(rng.integers(2, 33))
        Cin = 64 * int(rng.integers(1, 7))
        L = 2 * int(rng.integers(1, 2500))
        B = int(rng.integers(1, 9))
        relu, acc, want = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
        gy = (torch.randn(B, C, L, generator=g) * 1e-3).to(torch.bfloat16).to(DEV)
        raw = (torch.randn(B, C, L, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
        a, b, c0 = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV), (torch.randn(C, generator=g) * 1e-4).to(DEV)
        sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.4).to(DEV)
        W = (torch.randn(C, Cin, generator=g) * C ** -0.5).to(DEV)
        add = (torch.randn(B, Cin, L, generator=g) * 1e-3).to(torch.bfloat16).to(DEV) if acc else None
        wpt = ops.pointmlp_pack_transposed(W, 0, Cin, Cin, "bf16")
        one, zero = ops.const_vec(Cin, 1.0, DEV), ops.const_vec(Cin, 0.0, DEV)
        g_ref = ops.pointwise_bwd_apply(gy, raw, sc, sh, relu, a, b, c0)
        y_ref = ops.pointmlp(g_ref, wpt, one, zero, False, Cin, acc=add)
        y, gr = ops.pointmlp_bf16_bnb(gy, raw, wpt, one, zero, a, b, c0, sc, sh, relu, Cin, want_g_raw=want, acc=add)
        ok = eq(y, y_ref) and (gr is None or eq(gr, g_ref)) and ((gr is None) == (not want))
        bad += not ok
        print("bnb   B=%d C=%d Cin=%d L=%d relu=%d acc=%d g_raw=%d  %s" % (B, C, Cin, L, relu, acc, want, "ok" if ok else "MISMATCH"), flush=True)
        if case % 4:
            continue
        # ---- normalise-on-load on the streaming kernels (big launches only: every fourth case)
        K = int(rng.choice([64, 128, 256, 320]))
        C2 = int(rng.choice([0, 0, 64]))
        if C2 >= K:
            C2 = 0
        C1 = K - C2
        Cout = int(rng.choice([64, 128, 256, 384]))
        L = int(rng.integers(1100, 4000)) * 2
        gpc = (L + 63) // 64
        B = max(17, (8192 + gpc - 1) // gpc + int(rng.integers(0, 3)))
        if B * (K + Cout) * L * 2 > 5e9 or B * L * Cout * 4 < (32 << 20):
            continue
        r1 = (torch.randn(B, C1, L, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
        r2 = (torch.randn(B, C2, L, generator=g) * 1.5).to(torch.bfloat16).to(DEV) if C2 else None
        s1, h1 = (torch.rand(C1, generator=g) + 0.5).to(DEV), (torch.randn(C1, generator=g) * 0.4).to(DEV)
        s2, h2 = ((torch.rand(C2, generator=g) + 0.5).to(DEV), (torch.randn(C2, generator=g) * 0.4).to(DEV)) if C2 else (None, None)
        ra, rb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        W = (torch.randn(Cout, K, generator=g) * K ** -0.5).to(DEV)
        bias = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
        wp = ops.pointmlp_pack(W, "bf16")
        onec = ops.const_vec(Cout, 1.0, DEV)
        if not ops.bf16_xaff_ok(B, C1, C2, Cout, L):
            continue
        y1 = ops.channel_affine_act(r1, s1, h1, ra)
        y2 = ops.channel_affine_act(r2, s2, h2, rb) if C2 else None
        ref = ops.pointmlp_stats(y1, wp, onec, bias, False, Cout, x2=y2)
        got = ops.pointmlp_stats(r1, wp, onec, bias, False, Cout, x2=r2, xaff=(s1, h1, ra) + ((s2, h2, rb) if C2 else ()))
        ok = eq(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        if L % 8 == 0 and ops.wgrad_bf16_xaff_ok(B, Cout, C1, L):
            gg = (torch.randn(B, Cout, L, generator=g) * 1e-3).to(torch.bfloat16).to(DEV)
            ok = ok and torch.equal(ops.wgrad_bf16(gg, r1, xaff=(s1, h1, ra)), ops.wgrad_bf16(gg, y1))
        bad += not ok
        print("xaff  B=%d C1=%d C2=%d Cout=%d L=%d relu=%d/%d  %s" % (B, C1, C2, Cout, L, ra, rb, "ok" if ok else "MISMATCH"), flush=True)
print("fuzz_bf16_r6: %d mismatches" % bad)
sys.exit(1 if bad else 0)
