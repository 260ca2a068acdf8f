"""tools/grad_dev_h3.py -- end-to-end gradients of the f32-class training step against the float64 run of the reference (tests/golden/
train_step_*.npz: grad64/, ref32_dev/ = the deviation of the reference's own float32 run), for the variants of tools/ab_h3_train.py.
End-to-end gradients pass three arg-max pools: any change of rounding flips a few winners, so implementations of the same accuracy
scatter around the float32 reference's own deviation.  Prints rel-rms per parameter.

  python tools/grad_dev_h3.py [train_step_b8_n5000]"""
import os
import sys
from argparse import Namespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from models import networks as NW  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

DEV = "cuda:0"
fixture = sys.argv[1] if len(sys.argv) > 1 else "train_step_b8_n5000"
g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).contiguous()


def sub(t):
    f = t.detach().flatten()
    return f[::max(1, f.numel() // 16384)].cpu().numpy().astype(np.float64)


def rel_rms(a, r):
    return float(np.sqrt(np.mean((a - r) ** 2)) / np.sqrt(np.mean(r ** 2)))


keys = [k[7:] for k in g.files if k.startswith("grad64/") and not k.startswith("grad64/cls.")]
keys = [k for k in keys if np.sqrt(np.mean(g["grad64/" + k].astype(np.float64) ** 2)) >= 1e-5]
rows = {}
full = {}
for tag, seg, nol in (("A_store", False, False), ("B_sorted", True, False), ("C_sorted_nol", True, True)):
    ops.H3_SEGPOOL, ops.H3_NORM_ON_LOAD = seg, nol
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    enc.want_first_pn_out = False
    synth.fill_state_dict_(enc.state_dict(), seed)
    synth.fill_state_dict_(cls.state_dict(), seed + 1)
    enc.to(DEV).train()
    cls.to(DEV).train()
    with ops.precision("h3"):
        feat = enc(cu(g["pc"]), cu(g["sn"]), cu(g["node"]), cu(g["node_knn_I"]), is_train=True, epoch=0)
        loss = torch.nn.functional.cross_entropy(cls(feat, 0), cu(g["label"]))
        loss.backward()
    params = dict(enc.named_parameters())
    rows[tag] = {k: rel_rms(sub(params[k].grad), g["grad64/" + k].astype(np.float64)) for k in keys}
    full[tag] = {k: params[k].grad.detach().double().cpu().numpy() for k in keys}
    print("%s: loss %.8f (reference %.8f), feature max |d| vs reference = %.3g" % (tag, float(loss), float(g["loss"]), float(np.abs(feat.detach().cpu().numpy() - g["feature"]).max())))
print("%-44s %10s %10s %10s %10s   %10s %10s" % ("rel-rms vs the float64 run", "ref f32", "A", "B", "C", "B vs A", "C vs B"))
for k in keys:
    print("%-44s %10.2e %10.2e %10.2e %10.2e   %10.2e %10.2e" % (k, float(g["ref32_dev/" + k]), rows["A_store"][k], rows["B_sorted"][k], rows["C_sorted_nol"][k],
                                                              rel_rms(full["B_sorted"][k].ravel(), full["A_store"][k].ravel()), rel_rms(full["C_sorted_nol"][k].ravel(), full["B_sorted"][k].ravel())))
