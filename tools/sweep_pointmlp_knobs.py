import os
import sys as _sys
_sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs live in the variants build)
import sys
import torch
sys.path.insert(0, "/root/repo/so-net_amd")
from sonet_hip import ops
DEV = torch.device("cuda:0")
shapes = [(64, 393, 1024, 3072), (64, 1024, 512, 3072), (64, 320, 384, 15000), (64, 64, 128, 15000), (64, 128, 256, 15000), (64, 384, 320, 15000)]
for B, Cin, Cout, L in shapes:
    x = torch.randn(B, Cin, L, device=DEV); W = torch.randn(Cout, Cin, device=DEV) * 0.05
    sc = torch.ones(Cout, device=DEV); sh = torch.zeros(Cout, device=DEV)
    wp = ops.pointmlp_pack(W, "h3")
    for mt in ("0", "6", "4", "2"):
        for S in ("0", "2"):
            os.environ.pop("SONET_POINTMLP_MT", None); os.environ.pop("SONET_POINTMLP_S", None)
            if mt != "0": os.environ["SONET_POINTMLP_MT"] = mt
            if S != "0": os.environ["SONET_POINTMLP_S"] = S
            for _ in range(2): ops.pointmlp(x, wp, sc, sh, True, Cout)
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ops.pointmlp(x, wp, sc, sh, True, Cout)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            print("%4dx%-4d L=%-5d MT=%s S=%s : %8.1f us  %6.1f TF-eq" % (Cin, Cout, L, mt, S, ms * 1e3, 2.0 * Cin * Cout * B * L / ms / 1e9))
