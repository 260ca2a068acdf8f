"""Marginal cost of the stages of the headline forward with P graphs in flight (docs/findings.md R6.2).

The fused first PointNet owns every CU while it runs; SOM stage, node-level stage and head are short launches that run in what is left.
What does each of them add to the PERIOD of a batch when P graphs are replayed round-robin?  The same driver loop as bench.py with
stages cut out of the captured forward (the cut forwards compute garbage downstream: timing only):
    full            encoder + classifier head (the headline)
    no_head         encoder only
    no_node         SOM stage + fused first PointNet + head on a constant feature (node-level stage skipped)
    som_fused       SOM stage + fused first PointNet only
    fused_only      the fused kernel alone on a fixed grouping (no SOM stage): the floor
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "so-net_amd"), ROOT]
import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402
from sonet_hip.graph import GraphedForward  # noqa: E402

B, N, P = 64, 5000, int(os.environ.get("P", "3"))
STEPS = int(os.environ.get("STEPS", "300"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
opt = bench.make_opt(dev, B, N)


def build():
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    enc.want_first_pn_out = False
    synth.fill_state_dict_(enc.state_dict(), 0)
    synth.fill_state_dict_(cls.state_dict(), 1)
    return enc.to(dev).eval(), cls.to(dev).eval()


def variant(name):
    enc, cls = build()
    const_feat = torch.zeros((B, 1024), device=dev)
    if name in ("no_node", "som_fused"):
        enc._node_stage = lambda xp, prep, B_, M_: const_feat
    if name == "fused_only":
        state = {}
        real = enc.som_builder.assign_sort

        def cached(xd, snd, k, knn=None, **kw):
            if "r" not in state:
                state["r"] = real(xd, snd, k, knn=knn, **kw)
            return state["r"]
        enc.som_builder.assign_sort = cached
        enc._node_stage = lambda xp, prep, B_, M_: const_feat
    if name in ("full", "no_node"):
        return lambda pc, sn, node, knn: cls(enc(pc, sn, node, knn, is_train=False))
    return lambda pc, sn, node, knn: enc(pc, sn, node, knn, is_train=False)


def run(name):
    inps = [synth.make_inputs(B, N, seed=100 + 1000 * q, device=dev) for q in range(P)]
    fwds = [variant(name) for _ in range(P)]
    with torch.no_grad():
        graphs = [GraphedForward(f, (i_["pc"], i_["sn"], i_["node"], i_["node_knn_I"]), warmup=3) for f, i_ in zip(fwds, inps)]
    for g in graphs:
        g.check_every = 0
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]

    def many(n):
        for s in range(n):
            q = s % P
            with torch.cuda.stream(streams[q]):
                i_ = inps[q]
                graphs[q](i_["pc"], i_["sn"], i_["node"], i_["node_knn_I"])
    t_end = time.perf_counter() + 0.7
    while time.perf_counter() < t_end:
        many(P)
    res = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        many(STEPS)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / STEPS * 1e3)
    return res


if __name__ == "__main__":
    names = sys.argv[1:] or ["full", "no_head", "no_node", "som_fused", "fused_only", "full"]
    for n in names:
        r = run(n)
        print("P=%d %-11s ms per batch %s   (%.0f clouds/s)" % (P, n, " ".join("%.4f" % v for v in r), B / (sorted(r)[1] * 1e-3)), flush=True)
