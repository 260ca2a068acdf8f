"""tools/train_small_ops.py -- where the small aten launches of one bf16 training step come from: torch.profiler with Python stacks, the
callers of aten::zeros / zero_ / fill_ / copy_ / contiguous / sum / to, grouped by the innermost frame inside this repository."""
import collections
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
    synth.fill_state_dict_(enc.state_dict(), 0)
    synth.fill_state_dict_(cls.state_dict(), 1)
    enc.to(dev).train()
    cls.to(dev).train()
    inp = synth.make_inputs(B, N, seed=100, device=dev)
    oe = torch.optim.Adam(enc.parameters(), lr=1e-3)
    oc = torch.optim.Adam(cls.parameters(), lr=1e-3)

    def step():
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
        score = cls(feat, 0)
        enc.zero_grad(set_to_none=True)
        cls.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(score, inp["label"]).backward()
        oe.step()
        oc.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
    torch.cuda.synchronize()
want = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::contiguous", "aten::sum", "aten::to", "aten::clone", "aten::cat",
        "aten::empty_like", "aten::add", "aten::mul", "aten::where", "aten::full_like", "aten::ones_like", "aten::zeros_like", "aten::_to_copy")
cnt = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in want:
        st = list(ev.stack or [])
        fr = [f for f in st if "/so-net_amd/" in f or "bench.py" in f or "/tools/" in f]
        where = fr[0].split("/so-net_amd/")[-1] if fr else "(torch: %s)" % (st[0][-70:] if st else "no stack")
        cnt[(ev.key, where)] += ev.count
for (name, where), n in cnt.most_common(80):
    print("%3d  %-18s %s" % (n, name, where))
