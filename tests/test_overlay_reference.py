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


def test_layer_constructors_match_the_reference_for_every_option():
    """Every activation / normalisation option the reference's layer classes accept (models/layers.py:130-141,
    :176-187, :254-266) builds here with the same parameter / buffer names and shapes, the same sub-module names and
    the same activation module type; an option the reference does not know yields no norm / act there and here."""
    r = run("""
        import importlib.util, itertools
        from models import layers as ours
        from sonet_hip import overlay
        ref = overlay.reference_module("models", "layers.py", ours.__file__)
        def sig(m):
            return {"state": {k: list(v.shape) for k, v in m.state_dict().items()},
                    "children": {n: type(c).__name__ for n, c in m.named_children()}}
        out, bad = 0, []
        acts, norms = [None, "relu", "elu", "swish", "leakyrelu", "selu"], [None, "batch", "instance"]
        for act, norm in itertools.product(acts, norms):
            cases = {
                "EquivariantLayer": lambda L: L.EquivariantLayer(6, 16, activation=act, normalization=norm, momentum=0.2,
                                                                 bn_momentum_decay_step=3, bn_momentum_decay=0.5),
                "MyLinear": lambda L: L.MyLinear(8, 4, activation=act, normalization=norm, momentum=0.2,
                                                 bn_momentum_decay_step=3, bn_momentum_decay=0.5),
                "MyConv2d": lambda L: L.MyConv2d(3, 5, 1, activation=act, normalization=norm, momentum=0.2,
                                                 bn_momentum_decay_step=3, bn_momentum_decay=0.5),
                "UpConv": lambda L: L.UpConv(4, 2, activation=act, normalization=norm),
            }
            if act is not None:
                cases["PointNet"] = lambda L: L.PointNet(3, (8, 16), activation=act, normalization=norm)
                cases["PointResNet"] = lambda L: L.PointResNet(6, [8, 16, 32, 48], activation=act, normalization=norm)
                cases["KNNModule"] = lambda L: L.KNNModule(7, (8, 8), activation=act, normalization=norm)
            for name, make in cases.items():
                try:
                    a = sig(make(ref))
                except Exception as e:
                    a = "raises " + type(e).__name__
                try:
                    b = sig(make(ours))
                except Exception as e:
                    b = "raises " + type(e).__name__
                out += 1
                if a != b:
                    bad.append([name, act, norm, a, b])
        print(json.dumps({"n": out, "bad": bad[:5], "nbad": len(bad)}))
    """)
    assert r["n"] >= 100 and r["nbad"] == 0, r["bad"]


def test_network_constructors_match_the_reference_for_every_configuration():
    """Encoder / Classifier / Segmenter / Decoder variants (models/networks.py:71-109, :202-216, :230-257, :347-448):
    same state_dict keys and shapes as the reference's classes over the option combinations the task scripts use."""
    r = run("""
        import copy, itertools
        from models import networks as ours
        from sonet_hip import overlay
        import sys, types
        sys.modules.setdefault("faiss", types.ModuleType("faiss"))           # the reference's util/som.py imports it
        ref = overlay.reference_module("models", "networks.py", ours.__file__)
        sys.modules.pop("faiss", None)
        def keys(m):
            return {k: list(v.shape) for k, v in m.state_dict().items()}
        bad, n = [], 0
        for sn, som_k, node_num, fn, act, norm in itertools.product([True, False], [1, 9], [16, 64], [512, 1024],
                                                                      ["relu", "leakyrelu"], ["batch", "instance"]):
            o = copy.copy(opt)
            o.surface_normal, o.som_k, o.node_num, o.feature_num, o.activation, o.normalization = sn, som_k, node_num, fn, act, norm
            for name in ("Encoder", "Classifier", "Segmenter"):
                a, b = keys(getattr(ref, name)(o)), keys(getattr(ours, name)(o))
                n += 1
                if a != b:
                    bad.append([name, sn, som_k, node_num, fn, act, norm, sorted(set(a) ^ set(b))[:6]])
        for out_fc, conv_pc, conv_size, fn in itertools.product([256, 512], [1024, 4096], [32, 64], [512, 1024]):
            o = copy.copy(opt)
            o.output_fc_pc_num, o.output_conv_pc_num, o.output_conv_size, o.feature_num = out_fc, conv_pc, conv_size, fn
            o.output_pc_num = out_fc + conv_pc
            for name in ("DecoderLinear", "DecoderConv", "Decoder"):
                try:
                    a = keys(getattr(ref, name)(o))
                except Exception as e:
                    a = "raises " + type(e).__name__
                try:
                    b = keys(getattr(ours, name)(o))
                except Exception as e:
                    b = "raises " + type(e).__name__
                n += 1
                if a != b:
                    bad.append([name, out_fc, conv_pc, conv_size, fn, a if isinstance(a, str) else "keys", b if isinstance(b, str) else "keys"])
        print(json.dumps({"n": n, "nbad": len(bad), "bad": bad[:6]}))
    """)
    assert r["n"] >= 200 and r["nbad"] == 0, r["bad"]


@pytest.mark.timeout(900)
def test_committed_golden_fixtures_regenerate_bit_identically_from_the_live_reference(tmp_path):
    """oracle/make_golden.py, run against the mounted reference into a scratch directory, reproduces every committed
    tests/golden/*.npz array for array (inputs, integer outputs and float outputs alike): the fixtures the GPU box
    checks the HIP path against ARE outputs of the unmodified reference at this commit, not hand-kept numbers."""
    import glob
    import numpy as np
    env = dict(os.environ, SONET_GOLDEN_OUT=str(tmp_path))
    for extra in ([], ["autoencoder"], ["train5000"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")] + extra, env=env,
                           capture_output=True, text=True, timeout=800)
        assert p.returncode == 0, p.stderr[-3000:]
    made = sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    kept = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert made == kept, set(made) ^ set(kept)
    for name in kept:
        a, b = np.load(os.path.join(str(tmp_path), name)), np.load(os.path.join(ROOT, "tests", "golden", name))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind in "fc"), (name, k)
    assert json.load(open(os.path.join(str(tmp_path), "state_dict_keys.json"))) == \
        json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))


def test_task_script_imports_resolve_through_the_overlay():
    """What modelnet/train.py, shrec16/*.py, part-seg/train.py and autoencoder/train.py import besides the models
    (``util.util``, ``util.visualizer``, ``<task>.options``, ``util.potential_field``, ``data.augmentation``) comes from
    the reference checkout although ``util`` and ``models`` are overlay packages; BatchSOM.node_init reaches the reference's
    PotentialField the same way (util/som.py:196-206)."""
    r = run("""
        import importlib
        for m in ("dominate", "dominate.tags", "faiss"):
            sys.modules.setdefault(m, types.ModuleType(m))
        out = {}
        for name in ("util.util", "modelnet.options", "shrec16.options", "util.visualizer", "util.potential_field", "data.augmentation"):
            out[name] = importlib.import_module(name).__file__
        from util import som
        s = som.BatchSOM(2, 2, 3, 0, 3)
        s.node_init(3)                                   # 4 nodes: the potential-field initialiser runs in well under a second
        out["node_shape"] = list(s.node.shape)
        print(json.dumps(out))
    """)
    for name in ("util.util", "modelnet.options", "shrec16.options", "util.visualizer", "util.potential_field", "data.augmentation"):
        assert r[name].startswith(REF), (name, r[name])
    assert r["node_shape"] == [3, 3, 4]
