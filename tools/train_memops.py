"""tools/train_memops.py -- who launches the tiny copies and fills of one training step: every aten::copy_ / fill_ / zero_ (top-level or nested)
that reaches the device in one profiled step, with its shapes and the innermost Python frame inside this repository.

  python tools/train_memops.py [bf16|h3]"""
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()


def frame(stack):
    for s in stack:
        if ("so-net_amd" in s or "bench.py" in s) and "torch/" not in s:
            return s.replace(ROOT + "/", "")[:110]
    return (stack[0] if stack else "?")[:110]


cnt = collections.Counter()
allops = collections.defaultdict(lambda: [0, 0.0])
kern = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU:
        kern[e.name[:60]] += 1
        continue
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_") and e.device_time_total > 0:
        cnt[(e.name, str(e.input_shapes)[:60], frame(e.stack))] += 1
    # every OUTERMOST aten operator that reached the device (library kernels of the step: the at:: / Cijk_ / rocclr share of the profile)
    if e.name.startswith("aten::") and e.device_time_total > 0 and not (e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::")):
        allops[(e.name, str(e.input_shapes)[:70], frame(e.stack))][0] += 1
        allops[(e.name, str(e.input_shapes)[:70], frame(e.stack))][1] += e.device_time_total
for (name, shp, fr), n in sorted(cnt.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print("%3d  %-12s %-60s %s" % (n, name, shp, fr))
print(sum(cnt.values()), "device copies / fills per step")
print("device activities by name (top 12):")
for k, n in kern.most_common(12):
    print("%4d  %s" % (n, k))
print("outermost aten operators with device time (count, device us, op, shapes, frame):")
tot = 0.0
for (name, shp, fr), (n, us) in sorted(allops.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print("%3d %8.1f  %-28s %-70s %s" % (n, us, name, shp, fr))
print("%.1f us of device time in aten operators per step" % tot)
