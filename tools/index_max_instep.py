"""tools/index_max_instep.py MODE [ITERS] -- index_max_gather (64 x 384 x 15000 f32) alone and right behind the kernel that wrote its input.

In the training / segmenter step the kernel runs 20-26 % slower than stand-alone (0.33 vs 0.26 ms).  Modes (each prints the mean HIP-event
time of the index_max_gather launches; run each under rocprofv3 --pmc for the counters):
  alone         the same input tensor every launch, nothing in between
  after_writer  every launch right behind pointresnet_fused (store variant) writing that tensor (1.47 GB of fresh, dirty lines)
  after_other   every launch right behind the same writer filling ANOTHER tensor (same clock / power history, input not freshly written)
  after_idle    every launch after a host-side pause of 2 ms (clocks have dropped)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from models import layers as Lm  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

DEV = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, N, M = 64, 5000, 64
pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
synth.fill_state_dict_(pr.state_dict(), seed=7)
pr.to(DEV).eval()
inp = synth.make_inputs(B, N, seed=1, device=DEV)
a = ops.som_assign(inp["pc"], inp["node"], 3)
g = ops.som_group(inp["pc"], inp["sn"], a, want_augmented=True)
wstream, affine = pr._fused_state()
x = g["x_augmented"]
y = ops.pointresnet_fused(x, wstream, affine)
y_other = torch.empty_like(y)
row_max = g["row_max"]
ts = []
for it in range(iters + 2):
    if mode == "after_writer":
        y = ops.pointresnet_fused(x, wstream, affine)
    elif mode == "after_other":
        y_other = ops.pointresnet_fused(x, wstream, affine)
    elif mode == "after_idle":
        torch.cuda.synchronize()
        time.sleep(0.002)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.index_max_gather(y, a.min_idx_i32, M, row_max)
    e1.record()
    torch.cuda.synchronize()
    if it >= 2:
        ts.append(e0.elapsed_time(e1))
ms = sum(ts) / len(ts)
print("index_max_gather %-13s mean %.4f ms = %.2f TB/s of the algorithmic 1.485 GB  (min %.4f, max %.4f)" % (mode, ms, 1.485 / ms, min(ts), max(ts)))
