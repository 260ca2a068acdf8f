"""Overlay drop-in check (INTEGRATION.md level 1 / 2): the reference's own task classes
(models/classifier.py, segmenter.py, autoencoder.py -- unmodified, from the reference checkout) are built on
top of this repo's ``models.networks`` / ``layers`` / ``operations`` / ``losses`` / ``util.som`` / ``index_max``.

Runs only where a reference checkout is mounted (the build container); it never touches a GPU: construction,
checkpoint round trips, optimiser wiring, input staging, name resolution -- and that the forward pass fails
loudly instead of falling back to a CPU path.  Each case is its own process because the overlay and the
reference share top-level package names."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SONET_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "models", "classifier.py")),
                                reason="no reference checkout on this box")

PRELUDE = """
import json, sys, types
sys.path[:0] = [%r, %r]
for m in ("visdom", "h5py", "torchvision"):          # imported by the reference's task files, never called here
    sys.modules.setdefault(m, types.ModuleType(m))
from argparse import Namespace
import torch
opt = Namespace(gpu_id=0, device=torch.device("cpu"), batch_size=2, input_pc_num=256, surface_normal=True,
                feature_num=1024, activation="relu", normalization="batch", dropout=0.7, node_num=64, k=3, som_k=9,
                som_k_type="avg", bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40,
                lr=1e-3, pretrain=None, pretrain_lr_ratio=1, output_pc_num=1280, output_fc_pc_num=256,
                output_conv_pc_num=1024, output_conv_size=32, checkpoints_dir="/tmp")
""" % (os.path.join(ROOT, "so-net_amd"), REF)


def run(body):
    p = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_reference_task_models_build_on_the_overlay():
    r = run("""
        from models import classifier, segmenter, autoencoder, networks, layers, losses, operations
        from util import som
        import index_max
        out = {"files": {m.__name__: m.__file__ for m in (classifier, segmenter, autoencoder, networks, layers, losses, operations, som, index_max)}}
        for mod in (classifier, segmenter, autoencoder):
            m = mod.Model(opt)
            out[mod.__name__] = [type(m.encoder).__module__, type(m.encoder).__qualname__, type(m.encoder.som_builder).__module__,
                                 len(m.optimizer_encoder.param_groups[0]["params"])]
        print(json.dumps(out))
    """)
    ours, ref = os.path.join(ROOT, "so-net_amd"), REF
    for name in ("models.networks", "models.layers", "models.losses", "models.operations", "util.som", "index_max"):
        assert r["files"][name].startswith(ours), (name, r["files"][name])
    for name in ("models.classifier", "models.segmenter", "models.autoencoder"):
        assert r["files"][name].startswith(ref), (name, r["files"][name])
        assert r[name][:3] == ["models.networks", "Encoder", "util.som"]
        assert r[name][3] > 0


def test_reference_checkpoints_round_trip_through_the_overlay_models():
    """state_dict keys / shapes of Encoder + every head as the reference's own classes produce them (golden file
    written from the live reference), and save_network -> load_state_dict as modelnet/train.py:34 does."""
    r = run("""
        import os
        from models import classifier, segmenter, autoencoder
        m = classifier.Model(opt)
        keys = {"encoder": {k: list(v.shape) for k, v in m.encoder.state_dict().items()},
                "classifier": {k: list(v.shape) for k, v in m.classifier.state_dict().items()}}
        m.save_network(m.encoder, "encoder", "overlaytest", -1)
        path = os.path.join(opt.checkpoints_dir, "overlaytest_net_encoder.pth")
        m2 = classifier.Model(opt)
        missing = m2.encoder.load_state_dict(torch.load(path))
        os.remove(path)
        same = all(torch.equal(a, b) for a, b in zip(m.encoder.state_dict().values(), m2.encoder.state_dict().values()))
        print(json.dumps({"keys": keys, "same": same, "missing": [list(missing.missing_keys), list(missing.unexpected_keys)]}))
    """)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    assert r["keys"]["encoder"] == gold["encoder"] and r["keys"]["classifier"] == gold["classifier"]
    assert r["same"] and r["missing"] == [[], []]


def test_losses_names_off_the_hot_path_resolve_to_the_reference_file():
    r = run("""
        from models import losses
        import numpy as np
        out = {n: getattr(losses, n).__module__ for n in ("compute_iou", "compute_iou_np_array", "visualize_pc_seg",
                                                             "ChamferLoss", "CrossEntropyLossSeg", "robust_norm")}
        from util import som
        out["SOM"] = som.SOM.__module__
        out["BatchSOM"] = som.BatchSOM.__module__
        out["faiss_leaked"] = "faiss" in sys.modules
        try:
            losses.no_such_name
            out["missing"] = "no error"
        except AttributeError:
            out["missing"] = "AttributeError"
        # the per-point NLL of the segmenter head agrees with the reference class on CPU
        ours, theirs = losses.CrossEntropyLossSeg(), losses._reference_losses().CrossEntropyLossSeg()
        torch.manual_seed(0)
        s, t = torch.randn(2, 50, 64), torch.randint(0, 50, (2, 64))
        out["nll_equal"] = bool(torch.equal(ours(s, t), theirs(s, t)))
        v = torch.randn(2, 1, 3, 40)
        out["robust_norm_equal"] = bool(torch.equal(losses.robust_norm(v), losses._reference_losses().robust_norm(v)))
        print(json.dumps(out))
    """)
    for n in ("compute_iou", "compute_iou_np_array", "visualize_pc_seg"):
        assert r[n] == "models._reference_losses"
    for n in ("ChamferLoss", "CrossEntropyLossSeg", "robust_norm"):
        assert r[n] == "models.losses"
    assert r["SOM"] == "util._reference_som" and r["BatchSOM"] == "util.som"
    assert r["missing"] == "AttributeError" and not r["faiss_leaked"]
    assert r["nll_equal"] and r["robust_norm_equal"]


def test_overlay_forward_refuses_to_run_without_the_gpu():
    """set_input stages the batch as the reference does; forward must raise (no CPU fallback), not compute."""
    r = run("""
        from models import classifier
        from sonet_hip import synth
        m = classifier.Model(opt)
        inp = synth.make_inputs(2, 256, seed=1)
        m.set_input(inp["pc"], inp["sn"], inp["label"], inp["node"], inp["node_knn_I"])
        try:
            m.test_model()
            res = "computed"
        except RuntimeError as e:
            res = "RuntimeError"
        print(json.dumps({"res": res, "cuda": torch.cuda.is_available()}))
    """)
    if not r["cuda"]:
        assert r["res"] == "RuntimeError"
