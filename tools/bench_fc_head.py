"""tools/bench_fc_head.py -- the classifier head's training kernels (sonet_fc_*) at B = 64: 1024 -> 512 -> 256 -> 40, per launch (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for Cin, Cout, bn in ((1024, 512, True), (512, 256, True), (256, 40, False)):
    g = torch.Generator().manual_seed(Cin)
    x, W = torch.randn(B, Cin, generator=g).to(DEV), (torch.randn(Cout, Cin, generator=g) * 0.05).to(DEV)
    bias, gam, bet = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    gy = torch.randn(B, Cout, generator=g).to(DEV)

    def run():
        y, xh, inv = ops.fc_bn_act_fwd(x, W, bias, gam if bn else None, bet if bn else None, rm if bn else None, rv if bn else None, 0.1, 1e-5, bn)
        dz = ops.fc_bn_act_bwd(gy, y if bn else None, xh, inv, gam if bn else None, x, bn)[0]
        return ops.fc_dx(dz, W)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    with ops.kernel_timing() as rec:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    t = {}
    for n, e0, e1 in rec.records:
        t.setdefault(n, []).append(e0.elapsed_time(e1) * 1e3)
    print("B=%d %4d -> %3d: " % (B, Cin, Cout) + "   ".join("%s %.1f us" % (n.rsplit("_", 1)[0], sorted(v)[len(v) // 2]) for n, v in t.items()), flush=True)
