"""oracle/ref_harness.py -- TEST INFRASTRUCTURE: run the UNMODIFIED reference on CPU.

Works only where /root/reference is mounted (the build container); never imported on the GPU box
and never by the product.  Used by oracle/make_golden.py (fixture generation) and
oracle/ref_check.py (oracle-vs-live-reference differential check), both run as their own process
because the reference's top-level package names (``util``, ``models``, ``index_max``) are the very
names the drop-in product mirrors.

What is shimmed -- nothing in the reference tree is edited or copied:
* empty modules for third-party imports the classifier/segmenter path imports but never calls
  (faiss, torchvision, h5py, visdom: util/som.py:12, models/losses.py:9, data/*.py);
* a module ``index_max`` whose ``forward_cuda`` is the reference's own compiled ``forward_cpu``
  (oracle/_ref/index_max.so, built from models/index_max_ext/index_max.cpp by build_ref.py),
  because models/networks.py:182 hard-codes ``forward_cuda``;
* ``opt.gpu_id = 0`` (assert at util/som.py:187) with ``opt.device = cpu`` (fallback at :188);
* for the autoencoder only (``import_reference(with_faiss_shim=True)``): a ``faiss`` module whose ``IndexFlatL2`` is
  the exact brute-force search of oracle/sonet_oracle.c (faiss is un-vendored and not installed: the reference's
  ChamferLoss arithmetic -- gathers, robust_norm, means, models/losses.py:237-290 -- then runs unmodified, while
  the search itself is the restated one; parity is unpinned at the faiss boundary).
"""
import contextlib
import os
import sys
import types
from argparse import Namespace

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _faiss_shim():
    """Exact flat-L2 search standing in for faiss (autoencoder fixtures only)."""
    import numpy as np
    m = types.ModuleType("faiss")

    class StandardGpuResources(object):
        def setTempMemoryFraction(self, f):
            pass

    class GpuIndexFlatConfig(object):
        device = 0

    class IndexFlatL2(object):
        def __init__(self, d):
            assert d == 3
            self.db = None

        def add(self, x):
            self.db = np.ascontiguousarray(x, dtype=np.float32)

        def search(self, q, k):
            assert k == 1
            from oracle import cpu_oracle as _o
            q = np.ascontiguousarray(q, dtype=np.float32)
            idx = _o.chamfer_nn(np.ascontiguousarray(q.T[None]), np.ascontiguousarray(self.db.T[None]))[0]
            d = ((q - self.db[idx]) ** 2).sum(1, dtype=np.float32)
            return d[:, None], idx[:, None].astype(np.int64)

    m.StandardGpuResources, m.GpuIndexFlatConfig, m.IndexFlatL2 = StandardGpuResources, GpuIndexFlatConfig, IndexFlatL2
    m.index_cpu_to_gpu = lambda res, dev, index: index
    return m


def import_reference(with_faiss_shim=False):
    """Import the reference packages; returns a namespace of its modules."""
    import torch  # noqa: F401
    if not available():
        raise RuntimeError("/root/reference is not mounted")
    if with_faiss_shim:
        sys.modules["faiss"] = _faiss_shim()
    for name in ("faiss", "torchvision", "h5py", "visdom"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision.transforms"] = tv.transforms
    # reference's own index_max extension, CPU entry points
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import cpu_oracle as _o
    ref_ext = _o.ref_module()
    if ref_ext is None:
        from oracle import build_ref
        build_ref.build_ref()
        ref_ext = _o.ref_module()
    shim = types.ModuleType("index_max")
    shim.forward_cpu = ref_ext.forward_cpu
    shim.forward_multi_thread_cpu = ref_ext.forward_multi_thread_cpu
    shim.forward_cuda = ref_ext.forward_cpu            # networks.py:182 calls forward_cuda
    shim.forward_cuda_shared_mem = ref_ext.forward_cpu
    sys.modules["index_max"] = shim
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    som = importlib.import_module("util.som")
    layers = importlib.import_module("models.layers")
    operations = importlib.import_module("models.operations")
    networks = importlib.import_module("models.networks")
    classifier = importlib.import_module("models.classifier")
    segmenter = importlib.import_module("models.segmenter")
    autoencoder = importlib.import_module("models.autoencoder") if with_faiss_shim else None
    losses = importlib.import_module("models.losses")
    assert som.__file__.startswith(REF_ROOT) and networks.__file__.startswith(REF_ROOT)
    return Namespace(som=som, layers=layers, operations=operations, networks=networks, losses=losses,
                     classifier=classifier, segmenter=segmenter, autoencoder=autoencoder, index_max=shim)


def make_opt(**kw):
    """Option namespace with the defaults of modelnet/options.py:12-51 (hot-path relevant ones)."""
    import torch
    d = dict(gpu_id=0, device=torch.device("cpu"), batch_size=8, input_pc_num=5000, surface_normal=True,
             feature_num=1024, activation="relu", normalization="batch", lr=0.001, dropout=0.7,
             node_num=64, k=3, pretrain=None, pretrain_lr_ratio=1, som_k=9, som_k_type="avg",
             random_pc_dropout_lower_limit=1, bn_momentum=0.1, bn_momentum_decay_step=None,
             bn_momentum_decay=0.6, classes=40, checkpoints_dir="/tmp", name="oracle")
    d.update(kw)
    return Namespace(**d)


@contextlib.contextmanager
def sorted_topk():
    """Force ``sorted=True`` inside torch.topk while the reference runs.

    A sorted result is a legal result of ``topk(sorted=False)`` (util/som.py:253), so the reference
    is still the reference; its slot order then IS the canonical ascending order the HIP kernel
    emits, and min_idx / mask / index_max outputs compare bit-for-bit (hard part 1c).
    """
    import torch
    orig = torch.topk

    def patched(input, k, dim=-1, largest=True, sorted=True, **kwargs):
        return orig(input, k, dim=dim, largest=largest, sorted=True, **kwargs)

    torch.topk = patched
    try:
        yield
    finally:
        torch.topk = orig
