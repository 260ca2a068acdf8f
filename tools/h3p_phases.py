"""tools/h3p_phases.py -- where a workgroup of the third-generation layer spends its cycles (variants build: SONET_H3P_PROF phase
counters, SONET_H3P_ABL ablations).  Per (layer, variant): launch time, workgroups, mean cycles per workgroup in the prologue / the
chunk loops / the epilogues / in total (wave 0's clock), and the same with outputs dropped, X loads L1-hot, W requests L2-hot."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {"320x384": (64, 256, 384, 15000), "1024x512": (1024, 0, 512, 3072), "128x256": (128, 0, 256, 15000), "512x256": (512, 0, 256, 3072)}
B = 64
g = torch.Generator().manual_seed(0)
prof = torch.zeros(8, dtype=torch.int64, device=DEV)
os.environ["SONET_H3P_PROF"] = str(prof.data_ptr())
for name in sys.argv[1:] or ["320x384", "1024x512"]:
    C1, C2, Cout, L = SHAPES[name]
    x1 = ops.p16_from_f32(torch.randn(B, C1, L, generator=g).to(DEV))
    x2 = ops.p16_from_f32(torch.randn(B, C2, L, generator=g).to(DEV)) if C2 else None
    W = (torch.randn(Cout, C1 + C2, generator=g) * (2.0 / (C1 + C2)) ** 0.5).to(DEV)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    wp = ops.pointmlp_h3p_pack(W)
    print(name, flush=True)
    for shape, ns in (("4,2,2", None), ("6,2,1", 1), ("8,2,1", 1)):
        if (Cout // 32) % int(shape.split(",")[0]):
            continue
        for out in ("f32", "p16"):
            for abl in (0, 8, 15):
                os.environ["SONET_H3P_SHAPE"] = shape
                if ns:
                    os.environ["SONET_H3P_NSLAB"] = str(ns)
                else:
                    os.environ.pop("SONET_H3P_NSLAB", None)
                os.environ["SONET_H3P_ABL"] = str(abl)
                for _ in range(3):
                    ops.pointmlp_h3p(x1, wp, sc, sh, True, Cout, x2=x2, out=out)
                prof.zero_()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                n = 10
                for _ in range(n):
                    ops.pointmlp_h3p(x1, wp, sc, sh, True, Cout, x2=x2, out=out)
                b.record()
                torch.cuda.synchronize()
                p = prof.cpu().tolist()
                wgs = max(1, p[0])
                print("  shape %-7s slabs %-4s out %s abl %d: %.4f ms  %6d wgs/launch  cycles per workgroup: prologue %6.0f  loops %7.0f  epilogues %6.0f  total %7.0f"
                      % (shape, ns or "dflt", out, abl, a.elapsed_time(b) / n, wgs // n, p[1] / wgs, p[2] / wgs, p[3] / wgs, p[4] / wgs), flush=True)
