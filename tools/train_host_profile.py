"""tools/train_host_profile.py -- cProfile of the HOST side of the training step (20 steps, device drained before each so that no call
waits for the device): where the 4.9 ms of Python / ctypes / aten dispatch per step go.

  python tools/train_host_profile.py [bf16|h3] [n_rows]"""
import cProfile
import os
import pstats
import sys

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
    pr = cProfile.Profile()
    n = 20
    for _ in range(n):
        torch.cuda.synchronize()
        pr.enable()
        step()
        pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 45
print("per step (total / %d); tottime-sorted" % n)
tot = sum(v[2] for v in st.stats.values())
print("sum of tottime per step: %.3f ms" % (tot / n * 1e3))
items = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:rows]
for (fn, ln, name), (cc, nc, tt, ct, _) in items:
    print("%8.1f us tot %8.1f us cum  %5.1f calls  %s:%d %s" % (tt / n * 1e6, ct / n * 1e6, nc / n, fn.replace(ROOT + "/", "")[-50:], ln, name[:60]))
