"""tools/sweep_pointmlp.py -- timings of the layer-wise split kernel on the second-stage / segmenter shapes."""
import os
import sys as _sys
_sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs live in the variants build)
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "so-net_amd"))
from sonet_hip import ops
DEV = torch.device("cuda:0")
shapes = [(64, 515, 768, 64), (64, 768, 1024, 64), (64, 387, 512, 576), (64, 512, 512, 576), (64, 393, 1024, 3072), (64, 1024, 512, 3072),
          (64, 64, 128, 15000), (64, 320, 384, 15000)]
for B, Cin, Cout, L in shapes:
    x = torch.randn(B, Cin, L, device=DEV); W = torch.randn(Cout, Cin, device=DEV) * 0.05
    sc = torch.ones(Cout, device=DEV); sh = torch.zeros(Cout, device=DEV)
    for mode in ("h3", "x3"):
        wp = ops.pointmlp_pack(W, mode)
        for _ in range(3): ops.pointmlp(x, wp, sc, sh, True, Cout)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.pointmlp(x, wp, sc, sh, True, Cout)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%4dx%-4d L=%-5d %s : %8.1f us  %6.1f TF-eq" % (Cin, Cout, L, mode, ms * 1e3, 2.0 * Cin * Cout * B * L / ms / 1e9))
