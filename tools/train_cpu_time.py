"""tools/train_cpu_time.py -- how long the HOST needs to enqueue one training step (no synchronisation inside the loop) against the time the
device needs for it: host < device = the step is device-bound (the host runs ahead), host > device = launch-bound.

  python tools/train_cpu_time.py [bf16|h3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

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
    parts = {"forward": 0.0, "zero_grad + loss": 0.0, "backward": 0.0, "optimizers": 0.0}

    def step(acc):
        t0 = time.perf_counter()
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
        score = cls(feat, 0)
        t1 = time.perf_counter()
        enc.zero_grad(set_to_none=True)
        cls.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(score, inp["label"])
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        oe.step()
        oc.step()
        t4 = time.perf_counter()
        if acc:
            for k, d in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                parts[k] += d

    for _ in range(8):
        step(False)
    torch.cuda.synchronize()
    from sonet_hip import host
    host.freeze_gc()
    # (a) host only: one step at a time, the device drained before each (nothing to wait for, no queue limit)
    n = 10
    for _ in range(n):
        torch.cuda.synchronize()
        step(True)
    torch.cuda.synchronize()
    host = sum(parts.values()) / n
    print("host time to enqueue one step: %.3f ms  (%s)" % (host * 1e3, ", ".join("%s %.3f" % (k, v / n * 1e3) for k, v in parts.items())))
    # (b) device: back-to-back steps
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step(False)
    torch.cuda.synchronize()
    print("back-to-back: %.3f ms per step" % ((time.perf_counter() - t0) / n * 1e3))
