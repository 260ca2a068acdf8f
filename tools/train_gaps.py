"""tools/train_gaps.py -- where the GPU idles inside one training step: torch.profiler device activities of a step, the union of their
intervals over all streams, and the gaps longer than 3 us with the kernels on either side.

  python tools/train_gaps.py [bf16|h3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, N = 64, 5000
with ops.precision(sys.argv[1] if len(sys.argv) > 1 else "bf16"):
    opt = bench.make_opt(dev, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    enc.want_first_pn_out = False
    synth.fill_state_dict_(enc.state_dict(), 0)
    synth.fill_state_dict_(cls.state_dict(), 1)
    enc.to(dev).train()
    cls.to(dev).train()
    inp = synth.make_inputs(B, N, seed=100, device=dev)
    oe, oc = bench.make_adam(enc), bench.make_adam(cls)

    def step():
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
        score = cls(feat, 0)
        enc.zero_grad(set_to_none=True)
        cls.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(score, inp["label"]).backward()
        oe.step()
        oc.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()

ks = sorted([(e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type != torch.autograd.DeviceType.CPU],
            key=lambda k: k[0])
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy, cur_end, gaps, last = 0.0, ks[0][0], [], None
for s, e, n in ks:
    if s > cur_end:
        gaps.append((s - cur_end, last, n))
        busy += 0.0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last = n
print("3 steps: %.3f ms wall on the device, %.3f ms busy (union over streams), %.3f ms idle in %d gaps" % (
    (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, len(gaps)))
big = sorted(gaps, key=lambda g: -g[0])
print("gaps > 3 us: %d, their sum %.3f ms" % (sum(1 for g in gaps if g[0] > 3), sum(g[0] for g in gaps if g[0] > 3) / 1e3))
for d, a, b in big[:45]:
    print("%7.1f us   after %-58s before %s" % (d, (a or "")[:58], b[:70]))
