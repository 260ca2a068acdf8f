"""tools/one_h3p.py NAME [ITERS] -- launch one third-generation layer shape a few times (for rocprofv3 --pmc / --kernel-trace passes).
NAME: 320x384 | 1024x512 | 393x1024z | 512x256 | 128x256.  OUT=f32|p16 (default f32).  SONET_H3P_SHAPE / _NSLAB / _SWAP apply in the variants build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {"320x384": (64, 256, 384, 15000, False), "1024x512": (1024, 0, 512, 3072, False), "393x1024z": (384, 9, 1024, 3072, True),
          "512x256": (512, 0, 256, 3072, False), "128x256": (128, 0, 256, 15000, False)}
name = sys.argv[1] if len(sys.argv) > 1 else "320x384"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C1, C2, Cout, L, nodeadd = SHAPES[name]
B = int(os.environ.get("B", "64"))
g = torch.Generator().manual_seed(0)
x1 = ops.p16_from_f32(torch.randn(B, C1, L, generator=g).to(DEV))
x2 = ops.p16_from_f32(torch.randn(B, C2, L, generator=g).to(DEV)) if C2 else None
W = (torch.randn(Cout, C1 + C2, generator=g) * (2.0 / (C1 + C2)) ** 0.5).to(DEV)
sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
z = torch.randn(B, Cout, 64, generator=g).to(DEV) if nodeadd else None
zi = torch.randint(0, 64, (B, L), generator=g, dtype=torch.int32).to(DEV) if nodeadd else None
wp = ops.pointmlp_h3p_pack(W)
out = os.environ.get("OUT", "f32")
if os.environ.get("GEN2"):
    xf1, xf2 = x1.float(), (x2.float() if C2 else None)
    wp2 = ops.pointmlp_pack(W, "h3")
    for _ in range(iters):
        if nodeadd:
            ops.pointmlp_nodeadd(xf1, wp2, sc, sh, True, Cout, z, zi, x2=xf2)
        else:
            ops.pointmlp(xf1, wp2, sc, sh, True, Cout, x2=xf2)
else:
    for _ in range(iters):
        ops.pointmlp_h3p(x1, wp, sc, sh, True, Cout, x2=x2, z=z, zidx=zi, out=out)
torch.cuda.synchronize()
