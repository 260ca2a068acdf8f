"""tools/seg_kernels.py -- every launch of one segmenter forward (BASELINE configs[2], 64 x 1024 points) with its mean time (HIP events)."""
import os
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, N = 64, 1024
opt = bench.make_opt(dev, B, N)
opt.som_k_type = "center"
enc, seg = NW.Encoder(opt), NW.Segmenter(opt)
synth.fill_state_dict_(enc.state_dict(), 1)
synth.fill_state_dict_(seg.state_dict(), 2)
enc.to(dev).eval()
seg.to(dev).eval()
inp = synth.make_inputs(B, N, seed=3, device=dev)
label = torch.randint(0, 16, (B,), device=dev)
with torch.no_grad():
    for _ in range(3):
        NW.segmentation_forward(enc, seg, inp["pc"], inp["sn"], label, inp["node"], inp["node_knn_I"])
    torch.cuda.synchronize()
    with ops.kernel_timing() as rec:
        for _ in range(10):
            NW.segmentation_forward(enc, seg, inp["pc"], inp["sn"], label, inp["node"], inp["node_knn_I"])
        torch.cuda.synchronize()
tot = 0.0
for name, s_ in sorted(rec.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-44s %3d x %8.4f ms = %8.4f ms per step" % (name, s_["count"] // 10, s_["mean_ms"], s_["total_ms"] / 10))
    tot += s_["total_ms"] / 10
print("sum %.4f ms" % tot)
