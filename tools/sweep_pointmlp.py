import os, sys, torch, time
sys.path.insert(0, "/root/repo/so-net_amd")
from sonet_hip import ops
DEV = torch.device("cuda:0")
shapes = [(64, 515, 768, 64), (64, 768, 1024, 64), (64, 387, 512, 576), (64, 512, 512, 576)]
for B, Cin, Cout, L in shapes:
    x = torch.randn(B, Cin, L, device=DEV); W = torch.randn(Cout, Cin, device=DEV) * 0.05
    wp = ops.pointmlp_pack(W, "h3"); sc = torch.ones(Cout, device=DEV); sh = torch.zeros(Cout, device=DEV)
    for mt in ("6", "4", "2", "1"):
        for S in ("1", "2"):
            os.environ["SONET_POINTMLP_MT"] = mt; os.environ["SONET_POINTMLP_S"] = S
            for _ in range(3): ops.pointmlp(x, wp, sc, sh, True, Cout)
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.pointmlp(x, wp, sc, sh, True, Cout)
            e1.record(); torch.cuda.synchronize()
            print("%dx%d L=%d MT=%s S=%s : %.1f us" % (Cin, Cout, L, mt, S, e0.elapsed_time(e1) / 20 * 1e3))
