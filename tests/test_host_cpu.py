"""CPU-side tests (no GPU needed): the C-ABI library loads and exports every symbol the header
declares, the python mirror keeps the reference's module / state_dict surface, the product fails
loudly without a GPU (no CPU fallback), and nothing under so-net_amd/ touches oracle/."""
import ctypes
import json
import os
import re
import sys
from argparse import Namespace

import pytest
import torch

from conftest import GOLDEN, PKG, ROOT


def header_functions(variants=False):
    """Entry points the header declares: the product section, or (variants=True) the #ifdef SONET_VARIANTS section."""
    src = open(os.path.join(ROOT, "include", "sonet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"#ifdef SONET_VARIANTS(.*?)#endif", src, flags=re.S)
    var = m.group(1) if m else ""
    if variants:
        src = var
    elif m:
        src = src[:m.start()] + src[m.end():]
    return sorted(set(re.findall(r"\b(sonet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from sonet_hip import _lib
    names = header_functions()
    assert len(names) >= 15 and "sonet_index_max_f32" in names and "sonet_som_assign_f32" in names
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libsonet_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), "python binding and header disagree"
    # the variants-only entry points: declared under #ifdef SONET_VARIANTS, NOT exported by the product library, exported by the
    # variants library when it is built; and the product library reads no environment variable (no getenv import)
    vnames = header_functions(variants=True)
    assert set(vnames) == set(_lib.VARIANT_SIGNATURES) and vnames
    product = os.path.join(PKG, "lib", "libsonet_hip.so")
    rawp = ctypes.CDLL(product)
    for n in vnames:
        assert not hasattr(rawp, n), "the product library exports the variants-only %s" % n
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", product], stdout=subprocess.PIPE).stdout.decode()
    assert "getenv" not in syms, "the product library reads environment variables"
    if os.path.exists(_lib.VARIANTS_PATH):
        rawv = ctypes.CDLL(_lib.VARIANTS_PATH)
        for n in names + vnames:
            assert hasattr(rawv, n), "libsonet_hip_variants.so does not export %s" % n
    lib = _lib.load()
    assert lib.sonet_abi_version() == 1
    assert lib.sonet_build_arch() == b"gfx950"
    assert lib.sonet_pointmlp_pack_size(320, 384) == 12 * 40 * 256
    assert lib.sonet_pointmlp_pack_size(6, 64) == 2 * 1 * 256


def test_c_abi_argument_validation_without_gpu():
    """Status codes and messages for bad arguments are produced before any launch."""
    from sonet_hip import _lib
    lib = _lib.load()
    st = lib.sonet_index_max_f32(None, None, None, 1, 1, 1, 1, None)
    assert st == 1 and "NULL" in _lib.last_error()
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.sonet_index_max_f32(p, p, p, 1, 1, 4, 2000, None) == 2 and "1024" in _lib.last_error()
    assert lib.sonet_index_max_f32(p, p, p, 0, 1, 4, 4, None) == 1
    assert lib.sonet_som_assign_f32(p, p, 1, 4, 8, 9, p, None, p, p, None) == 1 and "k=9" in _lib.last_error()
    assert lib.sonet_pointmlp_f32(p, 6, p, 3, p, p, p, 1, p, 1, 8, 4, None) == 1       # C1 % 8 with a second input
    assert lib.sonet_som_group_f32(p, None, p, p, p, 1, 4, 8, 3, None, None, None, None, p, None) == 1  # x_aug needs sn


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_fails_loudly_without_gpu():
    import index_max
    from models import layers as L
    from sonet_hip import ops
    from sonet_hip._lib import SonetHipError
    from util import som
    data, index = torch.randn(1, 2, 8), torch.zeros(1, 8, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="CUDA"):
        index_max.forward_cuda(data, index, 4)
    with pytest.raises(SonetHipError, match="MI355X"):
        index_max.forward_cpu(data, index, 4)
    with pytest.raises(RuntimeError):
        L.EquivariantLayer(6, 8, "relu", "batch")(torch.randn(1, 6, 5))
    bs = som.BatchSOM(8, 8, 3, 0, 2)
    with pytest.raises(RuntimeError):
        bs.query_topk(torch.randn(2, 3, 10), 3)
    with pytest.raises(RuntimeError):
        ops.knn_gather(torch.randn(1, 3, 4), torch.zeros(1, 4, 2, dtype=torch.int64))


def test_product_never_touches_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        if os.path.basename(dirpath) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/|cpu_oracle|sonet_oracle", txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, "product files reference oracle/: %s" % bad


def test_state_dict_keys_match_the_reference():
    from models import layers as L
    from models import networks as NW
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    opt = Namespace(gpu_id=0, device=torch.device("cpu"), batch_size=2, input_pc_num=256, surface_normal=True,
                    feature_num=1024, activation="relu", normalization="batch", dropout=0.7, node_num=64, k=3, som_k=9,
                    som_k_type="avg", bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    mods = {"encoder": NW.Encoder(opt), "classifier": NW.Classifier(opt),
            "knnmodule": L.KNNModule(387, (512, 512), "relu", "batch"),
            "myconv2d": L.MyConv2d(4, 8, 1, activation="relu", normalization="batch")}
    for name, mod in mods.items():
        got = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert got == ref[name], name
    n_enc = sum(p.numel() for p in mods["encoder"].parameters())
    n_cls = sum(p.numel() for p in mods["classifier"].parameters())
    assert (n_enc, n_cls) == (1999041, 667944)


def test_reference_module_surface():
    import index_max
    from models import layers as L, operations
    from util import som
    for fn in ("forward_cpu", "forward_multi_thread_cpu", "forward_cuda", "forward_cuda_shared_mem"):
        assert callable(getattr(index_max, fn))                     # index_max.cpp:154-159
    for cls in ("Swish", "MyBatchNorm1d", "MyBatchNorm2d", "MyLinear", "MyConv2d", "UpConv", "EquivariantLayer",
                "KNNModule", "PointNet", "PointResNet"):
        assert hasattr(L, cls)
    assert callable(operations.knn_gather_wrapper) and callable(operations.knn_gather_by_indexing)
    bs = som.BatchSOM(8, 8, 3, 0, 4)
    assert tuple(bs.node.shape) == (4, 3, 64) and bs.node_num == 64 and tuple(bs.init_weighting_matrix.shape) == (64, 8, 8)
    for m in ("query_topk", "query", "batch_update", "optimize", "node_init", "get_weighting_matrix", "idx2multi"):
        assert callable(getattr(bs, m))
    with pytest.raises(AssertionError):
        som.BatchSOM(8, 8, 3, -1, 4)                                 # util/som.py:187


def test_bn_momentum_decay_rule():
    from models import layers as L
    bn = L.MyBatchNorm1d(4, momentum=0.5, momentum_decay_step=2, momentum_decay=0.6)
    bn.decay_momentum(None); assert bn.momentum == 0.5
    bn.decay_momentum(0); assert bn.momentum == 0.5
    bn.decay_momentum(5); assert abs(bn.momentum - 0.5 * 0.6 ** 2) < 1e-12
    bn.decay_momentum(100); assert bn.momentum == 0.01
    y = bn(torch.randn(8, 4, 5), epoch=3)                           # BN module itself is plain aten (CPU ok)
    assert y.shape == (8, 4, 5)


def test_synth_inputs_are_deterministic_and_shaped():
    from sonet_hip import synth
    a, b = synth.make_inputs(3, 100, seed=5), synth.make_inputs(3, 100, seed=5)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert a["pc"].shape == (3, 3, 100) and a["node"].shape == (3, 3, 64) and a["node_knn_I"].shape == (3, 64, 9)
    assert a["node_knn_I"].dtype == torch.int64 and torch.equal(a["node_knn_I"][:, :, 0], torch.arange(64).expand(3, 64))
    assert torch.allclose(a["sn"].norm(dim=1), torch.ones(3, 100), atol=1e-5)


def test_ctypes_signatures_match_the_header_prototypes():
    """Arity, argument kinds (pointer / int / long long / float / double) and return types of sonet_hip/_lib.py against
    the prototypes of include/sonet_hip.h: a drifted binding would pass garbage through ctypes without an error."""
    from sonet_hip import _lib
    src = open(os.path.join(ROOT, "include", "sonet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"\b(int|size_t|const\s+char\s*\*)\s*(sonet_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src)
    allsig = dict(_lib.SIGNATURES, **_lib.VARIANT_SIGNATURES)          # (the #ifdef SONET_VARIANTS section is parsed with the rest)
    assert len(protos) == len(allsig), (len(protos), len(allsig))

    def kind(param):
        p = " ".join(param.split())
        if p in ("void", ""):
            return None
        if "*" in p or "sonet_stream_t" in p:
            return ctypes.c_void_p
        base = p.rsplit(" ", 1)[0] if " " in p else p            # drop the parameter name
        return {"int": ctypes.c_int, "int32_t": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float,
                "double": ctypes.c_double, "size_t": ctypes.c_size_t}[base.replace("const ", "")]

    for ret, name, params in protos:
        want = [k for k in (kind(p) for p in params.split(",")) if k is not None]
        assert allsig[name] == want, (name, [t.__name__ for t in want], [t.__name__ for t in allsig[name]])
        want_ret = {"int": ctypes.c_int, "size_t": ctypes.c_size_t}.get(ret, ctypes.c_char_p)
        assert _lib._RESTYPES.get(name, ctypes.c_int) == want_ret, name


def test_fused_kernel_build_leaves_the_accumulation_registers_alone():
    """The fused first-PointNet kernel owns a[0:255] through inline asm; hipcc must never use them itself (a VGPR spilled into
    an 'unused' AGPR would be overwritten by the kernel's MFMAs), must not spill to scratch, and must not touch the VGPR
    destination of an asm MFMA while it is in flight.  tools/check_fused_asm.py compiles the source to ISA and checks."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_fused_asm
    kernels = check_fused_asm.audit(os.path.join(ROOT, "so-net_amd", "csrc", "pointresnet_fused.hip"), [])
    fused = {k: v for k, v in kernels.items() if "fused_kernel" in k}
    assert len(fused) == 3, list(kernels)          # store, store + P16 planes, pooled
    for name, k in fused.items():
        assert k["mfma"] == 1224, (name, k)
        assert k["accvgpr_by_compiler"] == 0 and k["scratch"] == 0 and k["early_reads"] == 0, (name, k)


def test_plain_attribute_fast_path_keeps_module_semantics():
    """models.layers._PlainAttrs sends re-assignments of plain attributes past nn.Module.__setattr__; parameters, buffers, sub-modules,
    first assignments and properties must behave exactly as on a plain nn.Module."""
    import torch.nn as nn
    from models.layers import EquivariantLayer, MyLinear, _PlainAttrs

    class M(_PlainAttrs, nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(2, 2)
            self.register_buffer("buf", torch.zeros(2))
            self.cache = None
            self._seen = []

        @property
        def prop(self):
            return self._seen

        @prop.setter
        def prop(self, v):
            self._seen = self._seen + [v]

    m = M()
    m.cache = torch.ones(3)                                        # plain attribute, re-assigned: the fast path
    assert "cache" in m.__dict__ and torch.equal(m.cache, torch.ones(3))
    m.cache = nn.Parameter(torch.ones(1))                          # a Parameter under an existing plain name: registered, as nn.Module does
    assert "cache" in dict(m.named_parameters()) and "cache" not in m.__dict__
    m.buf = torch.ones(2)                                          # a buffer stays a buffer
    assert "buf" in dict(m.named_buffers()) and torch.equal(m.state_dict()["buf"], torch.ones(2))
    m.lin = nn.Linear(3, 3)                                        # a sub-module stays registered
    assert dict(m.named_modules())["lin"].in_features == 3
    m.prop = 1
    m.prop = 2
    assert m.prop == [1, 2]                                        # property setters are honoured
    m.fresh = 5                                                    # first assignment: the normal path
    m.fresh = 6
    assert m.fresh == 6
    # the product modules carry the mixin and still expose the reference's parameters
    lay, fc = EquivariantLayer(4, 8, activation="relu", normalization="batch"), MyLinear(8, 4, activation="relu", normalization="batch")
    assert isinstance(lay, _PlainAttrs) and isinstance(fc, _PlainAttrs)
    assert sorted(k for k, _ in lay.named_parameters()) == ["conv.bias", "conv.weight", "norm.bias", "norm.weight"]
    lay._wp_key = ("a",)
    lay._wp_key = ("b",)
    assert lay._wp_key == ("b",) and "_wp_key" not in lay.state_dict()


def test_freeze_gc_freezes_and_unfreezes():
    import gc
    from sonet_hip import host
    before = gc.get_freeze_count()
    n = host.freeze_gc()
    assert n >= before and n > 1000 and gc.get_freeze_count() == n   # (a Python process tracks far more than a thousand objects)
    host.unfreeze_gc()
    assert gc.get_freeze_count() == 0


def test_encoder_keeps_first_pn_out_for_a_live_segmenter_unless_told_otherwise():
    """Training forwards do not write first_pn_out when only its per-node maximum is consumed -- unless a head reads it per point copy.
    A head that says so sets ``want_first_pn_out``; the reference's own segmenter Model does not, so a live Segmenter in the process
    keeps the tensor (models/segmenter.py:79-109 reads encoder.first_pn_out after the forward)."""
    import gc
    from models import networks as NW
    opt = Namespace(gpu_id=0, device=torch.device("cpu"), batch_size=2, input_pc_num=256, surface_normal=True,
                    feature_num=1024, activation="relu", normalization="batch", dropout=0.7, node_num=64, k=3, som_k=9,
                    som_k_type="avg", bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=50)
    gc.collect()
    base = len(NW._DENSE_HEADS)
    enc = NW.Encoder(opt)
    if base == 0:
        assert enc._wants_dense() is False
    seg = NW.Segmenter(opt)
    assert len(NW._DENSE_HEADS) == base + 1 and enc._wants_dense() is True
    enc.want_first_pn_out = False                                  # said explicitly: wins over the registry
    assert enc._wants_dense() is False
    enc.want_first_pn_out = True
    assert enc._wants_dense() is True
    del enc.__dict__["want_first_pn_out"]
    del seg
    gc.collect()
    assert len(NW._DENSE_HEADS) == base
    if base == 0:
        assert enc._wants_dense() is False


def test_backward_sums_hint_is_dropped_when_the_gradient_was_accumulated_into():
    """The BatchNorm-backward sums a dgrad launch leaves on its output (models.layers._bwd_sums_hint) are valid only for exactly that
    tensor: autograd adds a second consumer's gradient IN PLACE, which moves the version counter."""
    from models import layers as L
    raw = torch.zeros(2, 4, 8)
    gy = torch.ones(2, 4, 8)
    sums = torch.zeros(8, dtype=torch.float64)
    assert L._bwd_sums_hint(gy, raw, True) is None
    gy._sonet_bwd_sums = (sums, raw.data_ptr(), gy._version, True)
    assert L._bwd_sums_hint(gy, raw, True) is sums
    assert L._bwd_sums_hint(gy, raw, False) is None                # another mask
    assert L._bwd_sums_hint(gy, torch.zeros(2, 4, 8), True) is None   # another raw tensor
    gy.add_(1.0)                                                   # what autograd's accumulation does
    assert L._bwd_sums_hint(gy, raw, True) is None


def test_deferred_side_stream_join_is_inert_without_a_gpu():
    from sonet_hip import ops
    ops.join_side_streams()                                        # nothing pending: a no-op
    ss = ops.side_stream(torch.device("cpu"))
    with ss:
        pass
    ss.reads(torch.zeros(3))
    ss.join(defer=True)
    assert not ops._pending_join


def test_carried_gradient_is_consumed_once_and_only_in_the_pass_that_deposited_it():
    """models.layers._GradCarry (the gradient the pooled last layer hands to the second layer of the first PointNet outside autograd's
    accumulation) on a CPU miniature of that graph: y feeds a 'layer' and, with the layer's output, a 'pool'.  The full pass returns the
    sum of both gradients of y; a pass that runs the pool alone leaves a deposit nobody may use in a later pass."""
    from models import layers as L
    from sonet_hip import ops
    if ops._graph_task_id is None:
        pytest.skip("this torch has no graph-task query: the carry is never created")
    c = L._GradCarry()

    class Pool(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y, h):
            return y.sum() + 2.0 * h.sum()

        @staticmethod
        def backward(ctx, g):
            c.put(g * torch.ones(3))
            return None, 2.0 * g * torch.ones(3)

    class Layer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y):
            return 3.0 * y

        @staticmethod
        def backward(ctx, g):
            t = c.take()
            return 3.0 * g if t is None else 3.0 * g + t

    x = torch.zeros(3, requires_grad=True)
    y = x * 1.0
    h = Layer.apply(y)
    out = Pool.apply(y, h)
    (gx,) = torch.autograd.grad(out, x, retain_graph=True)
    assert torch.equal(gx, torch.full((3,), 7.0))                 # 1 (pool) + 3 * 2 (through the layer)
    assert c.g is None                                             # consumed
    (gx,) = torch.autograd.grad(out, x, retain_graph=True)
    assert torch.equal(gx, torch.full((3,), 7.0))                 # once per pass
    torch.autograd.grad(out, h, retain_graph=True)                 # the pool alone: a deposit is left behind
    assert c.g is not None
    (gx,) = torch.autograd.grad(h.sum(), x, retain_graph=True)     # the layer alone, another pass: the stale deposit is dropped
    assert torch.equal(gx, torch.full((3,), 3.0)) and c.g is None
    assert c.take() is None


def test_data_flow_knobs_default_to_what_was_measured_faster():
    """The round-5 data-flow switches of sonet_hip.ops: the measured-faster forms are on, the two measured-slower records (BatchNorm-backward
    sums in a producer's epilogue / on the sparse input gradient's store: docs/findings.md R5.9, R5.14) are off -- unless the environment
    says otherwise, which is how the A/B tools flip them."""
    from sonet_hip import ops
    on = ["H3_SEGPOOL", "H3_NORM_ON_LOAD", "TRAIN_ASSIGN_SORT", "POOLED_SIDE_STREAM", "DEFER_WGRAD_JOIN", "BNB_ON_LOAD", "GRAD_CARRY"]
    off = ["BWD_STATS_EPILOGUE", "POOLED_DGRAD_TAIL"]
    for name in on + off:
        env = "SONET_" + name
        if env in os.environ:
            continue
        assert getattr(ops, name) is (name in on), name
    assert not ops.pooled_dgrad_tail_ok(64, 256, torch.float32) or ops.POOLED_DGRAD_TAIL
    assert not ops.pooled_dgrad_tail_ok(64, 256, torch.bfloat16)


def test_host_model_of_the_entry_balanced_sparse_input_gradient():
    """The chunk logic of pooled_dgrad5_kernel (csrc/pointwise_bwd.hip) restated on the host for one channel: a round's sorted entries cut into
    sixteen chunks of equal length (the first n % 16 one longer, so the non-empty chunks are the first ones); a chunk whose first column began
    in the chunk before starts it from zero and parks the partial sum; after the round the first parked chunk of a run on one column adds the
    run to the accumulator in chunk order; a column that continues from the previous ROUND resumes from the accumulator.  Integer-valued
    entries: every order of additions gives the same float, so the model must reproduce the plain per-column sums exactly -- incl. rounds
    shorter than sixteen entries, chunks that are one column from end to end, and columns cut three times."""
    import numpy as np
    rng = np.random.default_rng(0)
    R, G = 24, 16                                                  # (the kernel: 384 entries per round; small here to force many rounds)

    def run(cols, vals, order):
        acc = {}
        for base in range(0, len(cols), R):
            n = min(R, len(cols) - base)
            ent = list(zip(cols[base:base + n], vals[base:base + n]))
            per, rem = divmod(n, G)
            span = [(g * per + min(g, rem), per + (1 if g < rem else 0)) for g in range(G)]
            headcol = [ent[lo][0] if (cnt > 0 and lo > 0 and ent[lo - 1][0] == ent[lo][0]) else -1 for lo, cnt in span]
            head = [0.0] * G
            for g in order(G):                                     # (the chunks run concurrently on the GPU: any order must do)
                lo, cnt = span[g]
                if cnt == 0:
                    continue
                in_head, cur = headcol[g] >= 0, ent[lo][0]
                s = 0.0 if in_head else acc.get(cur, 0.0)
                for c, v in ent[lo:lo + cnt]:
                    if c != cur:
                        if in_head:
                            head[g] = s
                        else:
                            acc[cur] = s
                        in_head, cur, s = False, c, acc.get(c, 0.0)
                    s += v
                if in_head:
                    head[g] = s
                else:
                    acc[cur] = s
            for g in range(1, G):
                hc = headcol[g]
                if hc >= 0 and headcol[g - 1] != hc:
                    s, k = acc.get(hc, 0.0), g
                    while k < G and headcol[k] == hc:
                        s += head[k]
                        k += 1
                    acc[hc] = s
        return acc

    for trial in range(600):
        n = int(rng.integers(1, 130))
        cols = np.sort(rng.integers(0, int(rng.integers(1, 12)), n)).tolist()
        vals = rng.integers(1, 100, n).astype(float).tolist()
        ref = {}
        for c, v in zip(cols, vals):
            ref[c] = ref.get(c, 0.0) + v
        assert run(cols, vals, range) == ref and run(cols, vals, lambda G: reversed(range(G))) == ref, (trial, cols)


def test_deferred_join_only_for_single_producer_hook_free_weights():
    """models/layers.py ``_grad_slot_empty``: the side-stream join of a layer's weight gradient may wait for the end of the
    backward pass only when nothing reads that gradient earlier -- the parameter has no .grad yet, no tensor / post-accumulate hook, and
    this node is its only producer among the graphs alive (tokens registered by the forwards, dead with their graph)."""
    import gc
    import weakref
    from models import layers as L
    p = torch.nn.Parameter(torch.zeros(4, 3))
    ref = weakref.ref(p)
    t1 = L._register_use(ref)
    assert L._grad_slot_empty(ref, t1)                       # one producer, nothing attached
    assert not L._grad_slot_empty(ref, t1)                   # the same node again (a second backward through a retained graph)
    del t1
    t2, t3 = L._register_use(ref), L._register_use(ref)      # two producers in one pass: the engine sums their gradients on the main stream
    assert not L._grad_slot_empty(ref, t2)                   # as soon as the second exists -- NEITHER may stay un-joined
    assert not L._grad_slot_empty(ref, t3)
    del t2, t3
    t4, t5 = L._register_use(ref), L._register_use(ref)
    del t4                                                   # a graph that was dropped without a backward
    gc.collect()
    assert L._grad_slot_empty(ref, t5)
    del t5
    t6 = L._register_use(ref)
    h = p.register_hook(lambda g: g)
    assert not L._grad_slot_empty(ref, t6)                   # a tensor hook reads the gradient inside the pass
    h.remove()
    del t6
    t7 = L._register_use(ref)
    h2 = p.register_post_accumulate_grad_hook(lambda q: None)
    assert not L._grad_slot_empty(ref, t7)
    h2.remove()
    del t7
    # ... unless the hook declares that it joins the side streams itself before it reads a gradient (sonet_hip.dp.GradientAllReducer's does)
    t7b = L._register_use(ref)
    joins = lambda q: None                                   # noqa: E731
    joins._sonet_joins_side_streams = True
    h3 = p.register_post_accumulate_grad_hook(joins)
    assert L._grad_slot_empty(ref, t7b)
    t7c = L._register_use(ref)
    h4 = p.register_post_accumulate_grad_hook(lambda q: None)   # one undeclared hook beside it: no deferral
    del t7b
    assert not L._grad_slot_empty(ref, t7c)
    h3.remove()
    h4.remove()
    del t7c
    t8 = L._register_use(ref)
    p.grad = torch.zeros_like(p)
    assert not L._grad_slot_empty(ref, t8)                   # accumulation into an existing .grad happens on the main stream
    assert not L._grad_slot_empty(ref, None) and not L._grad_slot_empty(None, t8)
