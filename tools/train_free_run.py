"""tools/train_free_run.py -- host timestamps of a free-running window of training steps as bench.py --mode train runs them (gradient
all-reducer + FusedAdam): per step, the host time of forward / backward / reduce / optimizers.  Where does the host block?

  python tools/train_free_run.py [bf16|h3] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import dp, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, N = 64, 5000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
with ops.precision(sys.argv[1] if len(sys.argv) > 1 else "bf16"):
    opt = bench.make_opt(dev, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), 0)
    synth.fill_state_dict_(cls.state_dict(), 1)
    enc.to(dev).train()
    cls.to(dev).train()
    inp = synth.make_inputs(B, N, seed=100, device=dev)
    use_red = os.environ.get("NO_REDUCER", "0") == "0"
    if use_red:
        dp.init_distributed(force=True)
        dp.broadcast_parameters([enc, cls])
    oe, oc = bench.make_adam(enc), bench.make_adam(cls)
    red = dp.GradientAllReducer([enc, cls], always_reduce=True) if use_red else None
    rows = []

    def step(log):
        t0 = time.perf_counter()
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
        score = cls(feat, 0)
        enc.zero_grad(set_to_none=True)
        cls.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(score, inp["label"])
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        if red is not None:
            red.reduce()
        t3 = time.perf_counter()
        oe.step()
        t4 = time.perf_counter()
        oc.step()
        t5 = time.perf_counter()
        if log:
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))

    bench._spin_up(lambda: step(False), 1.0, chunk=8)
    import gc
    gc_log, gc_t = [], [0.0]

    def _cb(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        elif info["generation"] >= 1:
            gc_log.append((info["generation"], (time.perf_counter() - gc_t[0]) * 1e3, info["collected"]))
    gc.callbacks.append(_cb)
    if os.environ.get("GC_FREEZE", "0") != "0":
        gc.collect()
        gc.freeze()
    for _ in range(8):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step(True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
print("reducer %s: %d free-running steps: %.3f ms per step (host finished enqueueing after %.3f ms per step)" % (use_red, K, t_all / K * 1e3, t_host / K * 1e3))
print("garbage collections of generation >= 1 during the run (generation, ms, collected):", [(g, round(ms, 1), c) for g, ms, c in gc_log][:12],
      "objects tracked:", len(gc.get_objects()), "frozen:", gc.get_freeze_count())
print("step:  forward  backward  reduce  opt_enc  opt_cls   (host ms)")
for i, r in enumerate(rows):
    if i < 16 or i % 8 == 0 or sum(r) > 1.3 * sum(rows[max(0, i - 1)]):
        print("%4d  %s   total %.2f" % (i, "  ".join("%6.2f" % (v * 1e3) for v in r), sum(r) * 1e3))
