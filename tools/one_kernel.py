"""tools/one_kernel.py <name> -- launch one hot kernel a few times (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "pointmlp"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(os.environ.get("B", "64"))
if name == "pointmlp":
    C1, C2, Cout, L = 64, 256, 384, 15000
    x1, x2 = torch.randn(B, C1, L, device=DEV), torch.randn(B, C2, L, device=DEV)
    wp = ops.pointmlp_pack(torch.randn(Cout, C1 + C2, device=DEV) * 0.08, os.environ.get("MODE", "h3"))
    sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
    y = torch.empty(B, Cout, L, device=DEV)
    for _ in range(iters):
        ops.pointmlp(x1, wp, sc, sh, True, Cout, x2=x2, out=y)
elif name == "fused":
    from models import layers as Lm
    from sonet_hip import synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    x = torch.randn(B, 6, 15000, device=DEV)
    with torch.no_grad():
        for _ in range(iters):
            pr(x)
elif name == "fused_pool":
    from models import layers as Lm
    from sonet_hip import synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    inp = synth.make_inputs(B, 5000, seed=1, device=DEV)
    a = ops.som_assign(inp["pc"], inp["node"], 3)
    sg = ops.som_sort_group(inp["pc"], inp["sn"], a)
    wstream, affine = pr._fused_state()
    for _ in range(iters):
        ops.pointresnet_fused_pool(sg, wstream, affine, 64)
elif name == "bf16_pool":
    from models import layers as Lm
    from sonet_hip import synth
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(DEV).eval()
    inp = synth.make_inputs(B, 5000, seed=1, device=DEV)
    a = ops.som_assign(inp["pc"], inp["node"], 3)
    sg = ops.som_sort_group(inp["pc"], inp["sn"], a)
    with ops.precision("bf16"):
        wstream, affine = pr._fused_state()
        for _ in range(iters):
            ops.pointresnet_bf16_pool(sg, wstream, affine, 64)
elif name == "index_max":
    data = torch.randn(B, 384, 15000, device=DEV)
    index = torch.randint(0, 64, (B, 15000), device=DEV, dtype=torch.int32)
    for _ in range(iters):
        ops.index_max(data, index, 64)
elif name == "som":
    from sonet_hip import synth
    inp = synth.make_inputs(B, 5000, seed=1, device=DEV)
    for _ in range(iters):
        a = ops.som_assign(inp["pc"], inp["node"], 3)
        ops.som_group(inp["pc"], inp["sn"], a, want_augmented=True)
torch.cuda.synchronize()
