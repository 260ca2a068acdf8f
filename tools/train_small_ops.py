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
    oe, oc = bench.make_adam(enc), bench.make_adam(cls)

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
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
    torch.cuda.synchronize()
want = ("aten::zero_", "aten::fill_", "aten::copy_")
cnt = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=20):
    if ev.key in want:
        st = list(ev.stack or [])
        fr = [f for f in st if any(k in f for k in ("layers.py", "networks.py", "ops.py", "operations.py", "dp.py", "optim.py", "som.py", "graph.py"))]
        leaf = " | ".join(x.split("/")[-1][-48:] for x in st[:3])
        where = (" <- ".join(f.split("/")[-1] for f in fr[:3]) if fr else "(no repo frame)") + "   [" + leaf + "]"
        cnt[(ev.key, where)] += ev.count
for (name, where), n in cnt.most_common(80):
    print("%3d  %-18s %s" % (n, name, where))
