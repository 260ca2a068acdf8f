"""tools/bf16_drift.py -- why the bf16 eval forward is further from the f32 oracle AFTER bench.py's timed training steps than before.

bench.py reports, for BASELINE configs[1] (bf16), the eval forward of the timed batch against the oracle twice: with the initial
weights (0.50 x the 5e-2 bound) and with the weights / BatchNorm running statistics the ~100 Adam steps on random labels left behind
(1.2-1.5 x).  This script repeats those steps and then, stage by stage (pooled first PointNet, KNN features, final PointNet, feature),
prints the bf16 and the f32-class (h3) error against the oracle with the SAME trained weights, next to the quantities that decide how
far a bf16 rounding error (2^-9 relative per stored value) is amplified: the folded BatchNorm scale gamma / sqrt(running_var + eps) of
every layer and the rms of each stage.  Run on the GPU box:  python tools/bf16_drift.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from models import networks as NW  # noqa: E402
from oracle import cpu_oracle as O  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, N, P = 64, 5000, 2
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 98


def stage_errors(enc, inp, precision):
    enc.eval()
    with torch.no_grad(), ops.precision(precision):
        enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=False)
    sd = {k: v.detach().float().cpu() for k, v in enc.state_dict().items()}
    ref = O.encoder_forward(sd, inp["pc"][:P].cpu(), inp["sn"][:P].cpu(), inp["node"][:P].cpu(), inp["node_knn_I"][:P].cpu(),
                            use_ref_index_max=O.ref_module() is not None)
    out = {}
    for name, got in (("first_pn_out_masked_max", enc.first_pn_out_masked_max), ("knn_feature_1", enc.knn_feature_1),
                      ("final_pn_out", enc.final_pn_out), ("feature", enc.feature)):
        b = ref[name].double().numpy()
        a = got[:P].detach().float().cpu().double().numpy()
        rms = float(np.sqrt(np.mean(b ** 2)))
        out[name] = (float((np.abs(a - b) / np.maximum(np.abs(b), rms)).max()), float(np.sqrt(np.mean((a - b) ** 2)) / rms), rms)
    ids_ok = bool(np.array_equal(enc.min_idx[:P].cpu().numpy(), ref["min_idx"]))
    return out, ids_ok


def bn_scales(enc):
    res = {}
    for name, m in enc.named_modules():
        if hasattr(m, "running_var") and m.running_var is not None and m.weight is not None:
            sc = (m.weight.detach() / torch.sqrt(m.running_var + m.eps)).abs()
            res[name] = (float(sc.max()), float(sc.mean()), float(m.running_var.min()))
    return res


def report(tag, enc, inp):
    print("== %s" % tag)
    for prec in ("bf16", "h3"):
        errs, ids_ok = stage_errors(enc, inp, prec)
        print("  %-4s node ids exact: %s" % (prec, ids_ok))
        for k, (worst, relrms, rms) in errs.items():
            print("       %-26s worst |err| / max(|ref|, rms) = %.3e   rel-rms = %.3e   rms(ref) = %.3g" % (k, worst, relrms, rms))
    sc = bn_scales(enc)
    print("  folded BatchNorm scale |gamma| / sqrt(running_var + eps)  (max, mean)  and  min running_var:")
    for k, (mx, mean, vmin) in sc.items():
        print("       %-34s %8.3f %8.3f   %.3e" % (k, mx, mean, vmin))


with ops.precision("bf16"):
    opt = bench.make_opt(dev, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), 0)
    synth.fill_state_dict_(cls.state_dict(), 1)
    enc.to(dev)
    cls.to(dev)
    inp = synth.make_inputs(B, N, seed=100, device=dev)
    report("initial weights", enc, inp)
    enc.train()
    cls.train()
    opt_e = torch.optim.Adam(enc.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt_c = torch.optim.Adam(cls.parameters(), lr=1e-3, betas=(0.9, 0.999))
    for _ in range(steps):
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0)
        score = cls(feat, 0)
        enc.zero_grad(set_to_none=True)
        cls.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(score, inp["label"])
        loss.backward()
        opt_e.step()
        opt_c.step()
    print("loss after %d steps on random labels: %.4f" % (steps, float(loss)))
    report("after %d Adam steps (bf16 training)" % steps, enc, inp)
