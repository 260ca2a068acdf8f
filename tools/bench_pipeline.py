"""tools/bench_pipeline.py [precision] -- the forward step with P HIP graphs in flight on P streams (independent batches of a serving
loop): does the node-level / SOM stage of one batch fill the CUs the other batch's kernels leave idle?"""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402
from sonet_hip.graph import GraphedForward  # noqa: E402

if len(sys.argv) > 1:
    ops.POINTMLP_PRECISION = sys.argv[1]
dev = torch.device("cuda", 0)
B, N, K = 64, 5000, 60
opt = bench.make_opt(dev, B, N)
enc, cls = NW.Encoder(opt), NW.Classifier(opt)
synth.fill_state_dict_(enc.state_dict(), 0)
synth.fill_state_dict_(cls.state_dict(), 1)
enc.to(dev).eval()
cls.to(dev).eval()
for P in (1, 2, 3):
    inps = [synth.make_inputs(B, N, seed=100 + p, device=dev) for p in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    graphs = []
    with torch.no_grad():
        for p in range(P):
            i = inps[p]
            graphs.append(GraphedForward(lambda pc, sn, node, knn: cls(enc(pc, sn, node, knn, is_train=False)),
                                         (i["pc"], i["sn"], i["node"], i["node_knn_I"]), warmup=3))
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for s in range(K):
                p = s % P
                with torch.cuda.stream(streams[p]):
                    i = inps[p]
                    out = graphs[p](i["pc"], i["sn"], i["node"], i["node_knn_I"])
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print("%s: %d graph(s) in flight: %.4f ms per step, %.1f clouds/s" % (ops.POINTMLP_PRECISION, P, el * 1e3 / K, B * K / el), flush=True)
        # the same outputs as a graph replayed alone
        ref = graphs[0](inps[0]["pc"], inps[0]["sn"], inps[0]["node"], inps[0]["node_knn_I"]).clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[0]):
            a = graphs[0](inps[0]["pc"], inps[0]["sn"], inps[0]["node"], inps[0]["node_knn_I"])
        if P > 1:
            with torch.cuda.stream(streams[1]):
                graphs[1](inps[1]["pc"], inps[1]["sn"], inps[1]["node"], inps[1]["node_knn_I"])
        torch.cuda.synchronize()
        print("   concurrent replay == solo replay:", torch.equal(a, ref))
    del graphs
