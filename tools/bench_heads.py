"""tools/bench_heads.py -- forward timings of the other heads on the same encoder (BASELINE configs 3 and 4):
part-segmentation (B x 50 x N scores) and autoencoder (decoder + multi-resolution Chamfer loss).  Not the headline
metric (bench.py); numbers for DESIGN.md."""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _variants  # noqa: E402,F401  (knobs / record kernels live in the variants build)
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from models import networks as NW, losses as LS  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

DEV = torch.device("cuda:0")


def opt_for(B, N, **kw):
    d = dict(gpu_id=0, device=DEV, batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024, activation="relu",
             normalization="batch", dropout=0.6, node_num=64, k=3, som_k=9, som_k_type="avg", bn_momentum=0.1,
             bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=50, output_fc_pc_num=256, output_conv_pc_num=1024)
    d.update(kw)
    return Namespace(**d)


def timeit(fn, steps=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    # three windows of `steps`, the best one: a window now and then contains a multi-millisecond stall of the eager path (an
    # allocator hipMalloc / hipFree when the big per-step tensors change size class), which is not what this line prices
    dt = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = min(dt, (time.perf_counter() - t0) / steps)
    with ops.kernel_timing() as rec:
        fn()
    torch.cuda.synchronize()
    top = sorted(rec.summary().items(), key=lambda kv: -kv[1]["total_ms"])[:int(os.environ.get("TOP", "6"))]
    return dt * 1e3, top


def main():
    for B, N in ((8, 1024), (64, 1024), (16, 5000)):
        opt = opt_for(B, N, som_k_type="center")
        enc, seg = NW.Encoder(opt), NW.Segmenter(opt)
        synth.fill_state_dict_(enc.state_dict(), 1)
        synth.fill_state_dict_(seg.state_dict(), 2)
        enc.to(DEV).eval(); seg.to(DEV).eval()
        inp = synth.make_inputs(B, N, seed=3, device=DEV)
        label = torch.randint(0, 16, (B,), device=DEV)
        with torch.no_grad():
            ms, top = timeit(lambda: NW.segmentation_forward(enc, seg, inp["pc"], inp["sn"], label, inp["node"], inp["node_knn_I"]))
        print("segmenter  B=%-3d N=%-5d : %8.3f ms/step  %9.0f clouds/s   top: %s" % (
            B, N, ms, B / ms * 1e3, ", ".join("%s %.2f" % (k, v["total_ms"]) for k, v in top)))
        if (B, N) == (64, 1024):
            # the same forward as three HIP graphs in flight on three streams (independent batches, as bench.py --in-flight 3)
            from sonet_hip.graph import GraphedForward
            P, K = int(os.environ.get("SONET_HEADS_INFLIGHT", "3")), 30
            with torch.no_grad():
                inps = [synth.make_inputs(B, N, seed=3 + q, device=DEV) for q in range(P)]
                graphs = [GraphedForward(lambda pc, sn, node, knn: NW.segmentation_forward(enc, seg, pc, sn, label, node, knn),
                                         (i["pc"], i["sn"], i["node"], i["node_knn_I"]), warmup=2) for i in inps]
                streams = [torch.cuda.Stream(device=DEV) for _ in range(P)]

                def many(n):
                    for s_ in range(n):
                        with torch.cuda.stream(streams[s_ % P]):
                            i = inps[s_ % P]
                            graphs[s_ % P](i["pc"], i["sn"], i["node"], i["node_knn_I"])
                many(2 * P)
                torch.cuda.synchronize()
                best = float("inf")
                for _ in range(3):
                    t0 = time.perf_counter()
                    many(K)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / K)
            print("segmenter  B=%-3d N=%-5d : %8.3f ms/step  %9.0f clouds/s   (%d HIP graphs in flight on %d streams)" % (B, N, best * 1e3, B / best, P, P))
            del graphs
    for B, N in ((8, 5000), (64, 5000)):
        opt = opt_for(B, N, classes=40)
        enc, dec, crit = NW.Encoder(opt), NW.Decoder(opt), LS.ChamferLoss(opt)
        synth.fill_state_dict_(enc.state_dict(), 1)
        synth.fill_state_dict_(dec.state_dict(), 2)
        enc.to(DEV).eval(); dec.to(DEV).eval()
        inp = synth.make_inputs(B, N, seed=3, device=DEV)

        def fwd():
            f = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], False, None)
            pred = dec(f)
            return crit(pred, inp["pc"]) + crit(dec.conv_pc4, inp["pc"])
        with torch.no_grad():
            ms, top = timeit(fwd)
        print("autoencoder B=%-3d N=%-5d : %8.3f ms/step  %9.0f clouds/s   top: %s" % (
            B, N, ms, B / ms * 1e3, ", ".join("%s %.2f" % (k, v["total_ms"]) for k, v in top)))


def chamfer_roofline():
    """configs[3]: 1280 predicted vs 5000 ground-truth points.  The search is VALU-bound: 8 exact f32 operations per pair for the
    distance (no FMA: bit-exact neighbours) + 3 for the row minimum + 3 for the column check = 14 lane-operations per pair in the
    single sweep; two one-direction launches need 2 x 11.  Roofline: 39.3 T lane-ops/s (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)."""
    peak = 39.3e12
    for B in (8, 64):
        g = torch.Generator().manual_seed(B)
        pred = (torch.rand(B, 3, 1280, generator=g) * 2 - 1).to(DEV)
        gt = (torch.rand(B, 3, 5000, generator=g) * 2 - 1).to(DEV)
        pairs = B * 1280 * 5000

        def t(fn, it=30):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / it
        ms2 = t(lambda: __import__("sonet_hip.variants", fromlist=["x"]).chamfer_nn2(pred, gt))
        ms1 = t(lambda: (ops.chamfer_nn(pred, gt), ops.chamfer_nn(gt, pred)))
        print("chamfer B=%-3d 1280 x 5000: one sweep %.4f ms = %.2f T lane-ops/s (%.3f of the VALU roof) | two launches %.4f ms (%.3f)"
              % (B, ms2, 14 * pairs / ms2 / 1e9, 14 * pairs / (ms2 * 1e-3) / peak, ms1, 22 * pairs / (ms1 * 1e-3) / peak), flush=True)


if __name__ == "__main__":
    if "chamfer" in sys.argv[1:]:
        chamfer_roofline()
    else:
        main()
        chamfer_roofline()
