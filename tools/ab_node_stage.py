"""tools/ab_node_stage.py -- the classifier forward (B = 64 x 5000 points, HIP-graph replay) with the round-4 node-level stage (A) and with
the flat stage of round 5 (B) in ONE process on ONE box, windows alternating A B A B ... : box-to-box and run-to-run clock differences
(+-5 % on the fused first PointNet alone) are larger than the difference the stage makes, so only an interleaved comparison says anything.

  python tools/ab_node_stage.py [--rounds 10] [--steps 40] [--in-flight 3]"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--in-flight", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=5000)
    args = ap.parse_args()
    import bench
    from models import networks as NW
    from sonet_hip import ops, synth
    from sonet_hip.graph import GraphedForward
    dev = torch.device("cuda:0")
    B, N, P = args.batch, args.points, args.in_flight
    opt = bench.make_opt(dev, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), 7)
    synth.fill_state_dict_(cls.state_dict(), 8)
    enc.to(dev).eval()
    cls.to(dev).eval()
    fwd = lambda pc, sn, node, knn: cls(enc(pc, sn, node, knn, is_train=False))      # noqa: E731
    inps = [synth.make_inputs(B, N, seed=100 + q, device=dev) for q in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    graphs = {}
    with torch.no_grad():
        for tag, flag in (("A_round4_stage", False), ("B_flat_stage", True)):
            ops.NODE_STAGE_P16 = flag
            graphs[tag] = [GraphedForward(fwd, (i_["pc"], i_["sn"], i_["node"], i_["node_knn_I"]), warmup=2) for i_ in inps]
    ops.NODE_STAGE_P16 = True
    fa, fb = graphs["A_round4_stage"][0].static_output, graphs["B_flat_stage"][0].static_output
    torch.cuda.synchronize()
    print("scores: max |A - B| = %.3g (max |A| = %.3g)" % (float((fa - fb).abs().max()), float(fa.abs().max())))

    def run(tag, n, p):
        for s_ in range(n):
            q = s_ % p
            with torch.cuda.stream(streams[q]):
                i_ = inps[q]
                graphs[tag][q](i_["pc"], i_["sn"], i_["node"], i_["node_knn_I"])

    def window(tag, p):
        run(tag, 2 * p, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(tag, args.steps, p)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / args.steps

    # spin up the clocks
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        run("A_round4_stage", P, P)
    torch.cuda.synchronize()
    res = {(t, p): [] for t in graphs for p in (1, P)}
    for r in range(args.rounds):
        for p in (1, P):
            for tag in graphs:
                res[(tag, p)].append(window(tag, p))
    for p in (1, P):
        a, b = res[("A_round4_stage", p)], res[("B_flat_stage", p)]
        ma, mb = statistics.median(a), statistics.median(b)
        print("in flight %d: A (round-4 stage) median %.4f ms [min %.4f]   B (flat stage) median %.4f ms [min %.4f]   A / B = %.4f   clouds/s B = %.0f"
              % (p, ma, min(a), mb, min(b), ma / mb, B / mb * 1e3))
        print("   A windows: " + " ".join("%.4f" % v for v in a))
        print("   B windows: " + " ".join("%.4f" % v for v in b))


if __name__ == "__main__":
    main()
