"""tools/layer_ops_trace.py -- every aten operator (name, input shapes) of ONE EquivariantLayer training forward + backward, in order."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from models import layers as LY  # noqa: E402
from sonet_hip import ops  # noqa: E402

dev = "cuda:0"
with ops.precision(sys.argv[1] if len(sys.argv) > 1 else "bf16"):
    lyr = LY.EquivariantLayer(128, 256, activation="relu", normalization="batch").to(dev).train()
    x = torch.randn(8, 128, 15000, device=dev).bfloat16().requires_grad_(True)
    for _ in range(2):
        y = lyr(x, None)
        y.float().sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        y = lyr(x, None)
        loss = y.float().sum()
        loss.backward()
    torch.cuda.synchronize()
evs = sorted([e for e in prof.events() if e.name.startswith("aten::")], key=lambda e: e.time_range.start)
top = []
for e in evs:
    if any(e.time_range.start >= t.time_range.start and e.time_range.end <= t.time_range.end and e is not t for t in evs):
        continue                                    # nested inside another aten op
    top.append(e)
for e in top:
    print("%-28s %s" % (e.name, str(e.input_shapes)[:110]))
print(len(top), "top-level aten ops")
