"""tools/ab_h3_train.py -- the f32-class ("h3") classifier training step (B = 64 x 5000 points: forward, backward, gradient all-reduce, FusedAdam --
what bench.py --mode train --precision h3 times) in the variants of round 5, in ONE process on ONE box, windows alternating A B C A B C ...:
    A  store + index_max        the last layer of the first PointNet writes first_pn_out, index_max pools it (round 4)
    B  sorted pool              the first PointNet on node-sorted columns, the last layer pools its own output (sonet_pointmlp_h3_segpool_f32)
    C  B + normalise-on-load    the hidden layers hand their RAW outputs on; no normalise + ReLU pass, no normalised activations in memory
    D  C, the sparse weight gradient of the pooled layer AFTER its sparse input gradient (not beside it on the side stream)
    E  C, no side stream anywhere in the backward
    F  C, the side stream joined at the END of the backward pass (weight gradients are not needed before) instead of at the end of every layer
    G  F, the BatchNorm / ReLU backward applied by the operand load of the input-gradient launch (which writes g_raw for the weight gradient)
    H  G, the two gradients of the first layer's output added by the store of the second layer's input-gradient launch (ops.GRAD_CARRY)
    I  H, the column-0 part and the BatchNorm-backward sums of the widest hidden layer on the store of the sparse input gradient (ops.POOLED_DGRAD_TAIL:
       off in the product, measured slower stand-alone -- docs/findings.md R5.14; not timed in a step)
then one instrumented step of each (per-kernel times by events on the launching stream).

  python tools/ab_h3_train.py [--rounds 6] [--steps 24] [--precision h3]"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
if "--only" not in sys.argv or "I" in sys.argv[sys.argv.index("--only") + 1].upper():
    import _variants  # noqa: E402,F401  (variant I rides on sonet_pooled_dgrad_tail_f32: a variants-build record since round 6)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--precision", default="h3")
    ap.add_argument("--only", default="", help="letters of the variants to run, e.g. ACD")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--all-kernels", action="store_true")
    args = ap.parse_args()
    import bench
    from models import networks as NW
    from sonet_hip import dp, host, ops, synth
    dev = torch.device("cuda:0")
    B, N = args.batch, args.points
    variants = [("A_store_index_max", False, False, True, True, False), ("B_sorted_pool", True, False, True, True, False),
                ("C_sorted_pool_norm_on_load", True, True, True, True, False),
                ("D_C_pooled_pair_in_sequence", True, True, False, True, False), ("E_C_no_side_stream_at_all", True, True, False, False, False),
                ("F_C_joins_at_the_end_of_backward", True, True, True, True, True),
                ("G_F_batchnorm_backward_on_load", True, True, True, True, True, True),
                ("H_G_carried_gradient", True, True, True, True, True, True, True),
                ("I_H_tail_of_the_sparse_input_gradient", True, True, True, True, True, True, True, True)]
    if args.only:
        variants = [v for v in variants if v[0][0] in args.only]
    with ops.precision(args.precision):
        opt = bench.make_opt(dev, B, N)
        enc, cls = NW.Encoder(opt), NW.Classifier(opt)
        enc.want_first_pn_out = False
        synth.fill_state_dict_(enc.state_dict(), 0)
        synth.fill_state_dict_(cls.state_dict(), 1)
        enc.to(dev).train()
        cls.to(dev).train()
        inp = synth.make_inputs(B, N, seed=100, device=dev)
        dp.init_distributed(force=True)
        dp.broadcast_parameters([enc, cls])
        oe, oc = bench.make_adam(enc), bench.make_adam(cls)
        red = dp.GradientAllReducer([enc, cls], always_reduce=True)

        def step():
            feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
            score = cls(feat, 0)
            enc.zero_grad(set_to_none=True)
            cls.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(score, inp["label"])
            loss.backward()
            red.reduce()
            oe.step()
            oc.step()
            return loss

        def select(v):
            ops.H3_SEGPOOL, ops.H3_NORM_ON_LOAD, ops.POOLED_SIDE_STREAM, ops.BWD_SIDE_STREAM, ops.DEFER_WGRAD_JOIN = v[1], v[2], v[3], v[4], v[5]
            ops.BNB_ON_LOAD = len(v) > 6 and v[6]
            ops.GRAD_CARRY = len(v) > 7 and v[7]
            ops.POOLED_DGRAD_TAIL = len(v) > 8 and v[8]

        def window(v):
            select(v)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / args.steps

        select(variants[0])
        bench._spin_up(step, 1.0, chunk=8)
        host.freeze_gc()
        res = {v[0]: [] for v in variants}
        for _ in range(args.rounds):
            for v in variants:
                res[v[0]].append(window(v))
        base = statistics.median(res[variants[0][0]])
        for v in variants:
            w = res[v[0]]
            print("%-28s median %.3f ms per step [min %.3f]  (A / this = %.3f)   windows: %s"
                  % (v[0], statistics.median(w), min(w), base / statistics.median(w), " ".join("%.3f" % x for x in w)))
        # one instrumented step of each variant (events around every C-ABI launch: the step runs serialised by them, the times are
        # the kernels' own)
        for v in ([] if args.no_kernels else variants):
            select(v)
            step()
            torch.cuda.synchronize()
            with ops.kernel_timing() as rec:
                step()
            torch.cuda.synchronize()
            summ = rec.summary()
            tot = sum(d["total_ms"] for d in summ.values())
            print("\n%s: %.3f ms in %d launches of the C ABI (aten / RCCL launches are not in this list)" % (v[0], tot, sum(d["count"] for d in summ.values())))
            for name, d in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[: (60 if args.all_kernels else 22)]:
                print("   %8.3f ms  x%-3d %s" % (d["total_ms"], d["count"], name))
        select(variants[-1])


if __name__ == "__main__":
    main()
