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
launching = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::sum", "aten::cat", "aten::add", "aten::add_",
             "aten::mul", "aten::mul_", "aten::where", "aten::to", "aten::_to_copy", "aten::full", "aten::full_like", "aten::ones_like", "aten::zeros_like",
             "aten::bmm", "aten::mm", "aten::matmul", "aten::addmm", "aten::gather", "aten::scatter_add_", "aten::index", "aten::sub", "aten::div", "aten::rsqrt",
             "aten::bitwise_not", "aten::lt", "aten::gt", "aten::eq", "aten::max", "aten::amax", "aten::mean", "aten::linear", "aten::batch_norm", "aten::dropout",
             "aten::relu", "aten::threshold_backward", "aten::native_batch_norm_backward", "aten::log_softmax", "aten::nll_loss_forward", "aten::transpose_copy")
evs = sorted([e for e in prof.events() if e.name.startswith("aten::")], key=lambda e: e.time_range.start)
top, end = [], -1
for e in evs:                                   # top-level operators only (nested ones are their implementation)
    if e.time_range.start >= end:
        top.append(e)
        end = e.time_range.end
cnt = collections.Counter((e.name, str(e.input_shapes)[:90]) for e in top if e.name in launching)
for (name, shp), n in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-22s %s" % (n, name, shp))
print(sum(cnt.values()), "launching aten operators per step")
