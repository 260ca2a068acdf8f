#!/usr/bin/env python3
"""Per-parameter deviation of the training step's gradients from a float64 run of the same step with the SAME arg-max routing
(tests/f64_classifier.py, the restatement pinned to the reference's float64 run), for each arithmetic and both training fixtures --
the table behind tests/test_gpu_parity.py::test_training_gradients_with_forced_routing.

    python tools/grad_forced_routing.py [--modes h3,x3,f32] [--fixtures train_step_b16_n512,train_step_b8_n5000] [--env KEY=VAL ...]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="h3,x3,f32")
ap.add_argument("--fixtures", default="train_step_b16_n512,train_step_b8_n5000")
ap.add_argument("--set", nargs="*", default=[], help="ops switches, e.g. DEFER_WGRAD_JOIN=0 BNB_ON_LOAD=0")
args = ap.parse_args()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from argparse import Namespace  # noqa: E402
from models import networks as NW  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

for kv in args.set:
    k, v = kv.split("=")
    setattr(ops, k, v not in ("0", "False", "false"))
    print("ops.%s = %s" % (k, getattr(ops, k)))
DEV = T.DEV
for fixture in args.fixtures.split(","):
    g = T.golden(fixture)
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    for mode in args.modes.split(","):
        opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                        activation="relu", normalization="batch", dropout=0.0, node_num=64, k=3, som_k=9, som_k_type="avg",
                        bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
        enc, cls = NW.Encoder(opt), NW.Classifier(opt)
        synth.fill_state_dict_(enc.state_dict(), seed)
        synth.fill_state_dict_(cls.state_dict(), seed + 1)
        enc.to(DEV).train()
        cls.to(DEV).train()
        enc.want_first_pn_out = False
        with ops.precision(mode):
            cap = T._capture_stage(enc)
            feat = enc(T.cu(g["pc"]), T.cu(g["sn"]), T.cu(g["node"]), T.cu(g["node_knn_I"]), is_train=True, epoch=0)
            score = cls(feat, 0)
            cap.update(T._routing_of(enc, feat))
            loss = torch.nn.functional.cross_entropy(score, T.cu(g["label"]))
            cap["masks"] = T._relu_masks_of(loss, enc, cls)
            print("masks:", {k: tuple(v.shape) for k, v in cap["masks"].items()})
            loss.backward()
        torch.cuda.synchronize()
        r = T._f64_step(enc, cls, g, cap, forced=True)
        free = T._f64_step(enc, cls, g, cap, forced=False)
        fl = [int((free["route"][p] != cap[p]).sum()) for p in ("pool1", "pool2", "pool3")]
        mine = {k: p.grad for k, p in enc.named_parameters() if p.grad is not None}
        mine.update({"cls." + k: p.grad for k, p in cls.named_parameters() if p.grad is not None})
        print("\n== %s  %s  sorted=%s  loss %.9f (f64 forced %.9f)  flips vs free f64 run: pool1 %d pool2 %d pool3 %d"
              % (fixture, mode, cap["pos0"] is not None, float(loss), float(r["loss"]), fl[0], fl[1], fl[2]))
        rows = []
        for k, ref in r["grads"].items():
            rn = float(ref.norm()) / max(1.0, float(ref.numel()) ** 0.5)
            if rn < 1e-7 or k not in mine:
                continue
            rows.append((float((mine[k].double() - ref).norm() / ref.norm()), k))
        for rel, k in sorted(rows, reverse=True)[:12]:
            print("   %-45s %.3e" % (k, rel))
