"""tools/first_process.py -- the first GPU process on a fresh box runs the same training step at 7.6 ms where later processes need 5.3 ms.
Windows of back-to-back steps over ~8 s in ONE process (does it recover inside the process?), to be run twice in one gpurun call."""
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
with ops.precision("bf16"):
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

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    out = []
    for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 10)
    print("ms per step, windows of 100 steps:", " ".join("%.2f" % v for v in out), flush=True)
