"""Tensor-level wrappers over the C ABI: validate, allocate outputs with torch, launch on the
current stream.  Everything here requires CUDA (ROCm) tensors on an MI355X and raises otherwise
(mirroring the reference's CHECK_CUDA / CHECK_CONTIGUOUS -> RuntimeError convention,
models/index_max_ext/index_max.cpp:119-121)."""
import torch

from . import _lib
from ._lib import SonetHipError, check, ptr, stream_ptr
import functools as _functools
import threading as _threading

# per-THREAD host state: the C side's BatchNorm rider is thread_local, so is the record of having armed it.  (The deferred side-stream
# joins are NOT: autograd runs backward nodes on its per-device worker thread and the end-of-pass callback on whichever thread finished
# the pass -- they share one locked table, and joining a stream somebody else registered is only a wait.)
_tls = _threading.local()
_join_lock = _threading.Lock()


def _consumes_rider(fn):
    """A statistics-producing call: whatever happens inside it -- a failed argument check included -- the BatchNorm rider armed for it
    (``bn_rider``) is disarmed when it returns, so that it can never ride on an unrelated later launch."""
    @_functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        finally:
            _rider_done()
    return wrapped


# ---- optional per-launch timing with HIP events on the launch stream (used by bench.py) ----------
_TIMING = None


class kernel_timing:
    """``with ops.kernel_timing() as rec:`` records one (name, start, end) event pair per C-ABI launch
    on the current stream; ``rec.summary()`` (after a synchronize) gives per-name count and mean ms."""

    def __enter__(self):
        global _TIMING
        self.records = []
        _TIMING = self.records
        return self

    def __exit__(self, *exc):
        global _TIMING
        _TIMING = None
        return False

    def summary(self):
        out = {}
        for name, e0, e1 in self.records:
            d = out.setdefault(name, [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        return {k: dict(count=v[0], total_ms=v[1], mean_ms=v[1] / v[0]) for k, v in out.items()}


class _timed:
    __slots__ = ("name", "e0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _TIMING is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _TIMING.append((self.name, self.e0, e1))
        return False


def _chk(t, name, dtype=None, dim=None):
    if not isinstance(t, torch.Tensor):
        raise SonetHipError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise SonetHipError("%s must be a CUDA tensor/variable" % name)          # CHECK_CUDA
    if not t.is_contiguous():
        raise SonetHipError("%s must be contiguous" % name)                      # CHECK_CONTIGUOUS
    if dtype is not None and t.dtype != dtype:
        raise SonetHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if dim is not None and t.dim() != dim:
        raise SonetHipError("%s must be %d-D, got shape %s" % (name, dim, tuple(t.shape)))


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t is not None and t.device != dev:
            raise SonetHipError("tensors are on different devices: %s vs %s" % (dev, t.device))
    _lib.require_device(dev)
    return dev


# ------------------------------------------------------------------------------------------ index_max
def index_max(data, index, K):
    """data BxCxN' f32|bf16, index BxN' i32 -> BxCxK i32 (include/sonet_hip.h: sonet_index_max_*)."""
    _chk(data, "data", dim=3)
    _chk(index, "index", torch.int32, 2)
    if data.dtype not in (torch.float32, torch.bfloat16):
        raise SonetHipError("data must be float32 or bfloat16, got %s" % data.dtype)
    B, C, Np = data.shape
    if tuple(index.shape) != (B, Np):
        raise SonetHipError("index must be B x N' = %s, got %s" % ((B, Np), tuple(index.shape)))
    dev = _same_device(data, index)
    out = torch.empty((B, C, int(K)), dtype=torch.int32, device=dev)
    if out.numel() == 0 or Np == 0:
        return out.zero_()
    lib = _lib.load()
    with _lib.on_device(dev), _timed("index_max"):
        fn = lib.sonet_index_max_f32 if data.dtype == torch.float32 else lib.sonet_index_max_bf16
        check(fn(ptr(data), ptr(index), ptr(out), B, C, Np, int(K), stream_ptr()), "sonet_index_max")
    return out


def index_max_gather(data, index, K, row_max=None):
    """index_max + the masked gather of models/networks.py:185 in one pass -> (idx i32, val f32) BxCxK."""
    _chk(data, "data", dim=3)
    if data.dtype not in (torch.float32, torch.bfloat16):
        raise SonetHipError("data must be float32 or bfloat16, got %s" % data.dtype)
    _chk(index, "index", torch.int32, 2)
    B, C, Np = data.shape
    if tuple(index.shape) != (B, Np):
        raise SonetHipError("index must be B x N' = %s, got %s" % ((B, Np), tuple(index.shape)))
    if row_max is not None:
        _chk(row_max, "row_max", torch.int32, 2)
    dev = _same_device(data, index, row_max)
    idx = torch.empty((B, C, int(K)), dtype=torch.int32, device=dev)
    val = torch.empty((B, C, int(K)), dtype=torch.float32, device=dev)
    lib = _lib.load()
    fn = lib.sonet_index_max_gather_f32 if data.dtype == torch.float32 else lib.sonet_index_max_gather_bf16
    with _lib.on_device(dev), _timed("index_max_gather" if data.dtype == torch.float32 else "index_max_gather_bf16"):
        check(fn(ptr(data), ptr(index), ptr(row_max), ptr(idx), ptr(val), B, C, Np, int(K), stream_ptr()), "sonet_index_max_gather")
    return idx, val


def pointmlp_bf16_pool_ok(x1, x2, Cout, M):
    """The shapes ``pointmlp_bf16_pool`` takes (the rest raise SONET_ERR_UNSUPPORTED): bf16 CUDA inputs, even L, (C1 + C2) % 64 == 0,
    C1 % 16 == 0 with a second input, Cout % 32 == 0, M <= 255, and the LDS budget of a (cloud, slab) workgroup."""
    if not (POOLED_TRAIN_EPILOGUE and x1.is_cuda and x1.dtype == torch.bfloat16 and (x2 is None or x2.dtype == torch.bfloat16)):
        return False
    B, C1, L = x1.shape
    C2 = x2.shape[1] if x2 is not None else 0
    if L % 2 or L > 65534 or (C1 + C2) % 64 or (C2 and C1 % 16) or Cout % 32 or not 0 < M <= 255 or B > 65535:
        return False
    KC, CT = (C1 + C2) // 16, Cout // 32
    for ns in range(1, CT + 1):                       # (the launcher's slab search: any slab whose W rows + bins + ids fit 158 KiB)
        if CT % ns:
            continue
        tps = CT // ns
        if tps % 2 and tps % 3:
            continue
        if tps * KC * 1024 + tps * 32 * 8 + tps * 32 * M * 8 + tps * 32 * 4 + ((L + 15) & ~15) <= 158 * 1024:
            return True
    return False


def pointmlp_bf16_pool(x1, wp, scale, shift, relu, Cout, ids, M, row_max=None, x2=None, xaff=None):
    """The bf16 layer and the per-node arg-max pool of its output in one launch (``sonet_pointmlp_bf16_pool``); the B x Cout x L output is
    never written.  -> (idx i32, val f32) B x Cout x M: exactly ``index_max_gather(pointmlp(...), ids, M, row_max)``.
    xaff = (s1, h1, relu1[, s2, h2, relu2]): x1 / x2 are RAW outputs of BatchNorm layers, normalised by the operand load."""
    _chk(x1, "x", torch.bfloat16, 3)
    B, C1, L = x1.shape
    C2 = 0
    if x2 is not None:
        _chk(x2, "x2", torch.bfloat16, 3)
        if x2.shape[0] != B or x2.shape[2] != L:
            raise SonetHipError("x2 must be B x C2 x L")
        C2 = x2.shape[1]
    if wp.dtype != torch.int16:
        raise SonetHipError("pointmlp_bf16_pool: a bf16 pack")
    _chk(ids, "ids", torch.int32, 2)
    if tuple(ids.shape) != (B, L):
        raise SonetHipError("ids must be B x L")
    if row_max is not None:
        _chk(row_max, "row_max", torch.int32, 2)
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    dev = _same_device(x1, x2, wp, scale, shift, ids, row_max)
    lib = _lib.load()
    if wp.numel() != lib.sonet_pointmlp_bf16_pack_size(C1 + C2, Cout) // 2:
        raise SonetHipError("packed weight does not match Cin=%d Cout=%d" % (C1 + C2, Cout))
    idx = torch.empty((B, Cout, int(M)), dtype=torch.int32, device=dev)
    val = torch.empty((B, Cout, int(M)), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("pointmlpbf16_pool_%dx%d_L%d%s" % (C1 + C2, Cout, L, "_xaff" if xaff is not None else "")):
        if xaff is not None:
            check(lib.sonet_pointmlp_bf16_pool_xaff(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(ids), ptr(row_max),
                                                    ptr(idx), ptr(val), B, Cout, L, int(M), *_xaff_args(xaff, C1, C2, dev), stream_ptr()),
                  "sonet_pointmlp_bf16_pool_xaff")
        else:
            check(lib.sonet_pointmlp_bf16_pool(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(ids), ptr(row_max),
                                               ptr(idx), ptr(val), B, Cout, L, int(M), stream_ptr()), "sonet_pointmlp_bf16_pool")
    return idx, val


def pointmlp_h3_segpool_ok(x1, x2, wp, Cout, M):
    """The shapes ``pointmlp_h3_segpool`` takes: f32 CUDA inputs, an h3 pack, Cout % 32 == 0, C1 % 16 == 0 with a second input."""
    if not (POOLED_TRAIN_EPILOGUE and H3_SEGPOOL and x1.is_cuda and x1.dtype == torch.float32 and (x2 is None or x2.dtype == torch.float32)
            and wp.dtype == torch.int8):
        return False
    return Cout % 32 == 0 and (x2 is None or x1.shape[1] % 16 == 0) and 0 < M <= 1024


def pointmlp_h3_segpool(x1, wp, scale, shift, relu, Cout, ids_sorted, pos0, M, row_max=None, x2=None, xaff=None):
    """The fp16-split layer and the per-node arg-max pool of its output in one pass over NODE-SORTED columns (``sonet_pointmlp_h3_segpool_f32``);
    the B x Cout x L output is never written.  ids_sorted B x L i32 (``som_sort_group``), pos0 B i32 = sorted position of original column 0.
    -> (idx i32, val f32) B x Cout x M: ``index_max_gather(pointmlp(...), ids_sorted, M, row_max)`` with pos0[b] (and the value there) in
    place of position 0 for bins nothing beat and for masked nodes."""
    _chk(x1, "x", torch.float32, 3)
    B, C1, L = x1.shape
    C2 = 0
    if x2 is not None:
        _chk(x2, "x2", torch.float32, 3)
        if x2.shape[0] != B or x2.shape[2] != L:
            raise SonetHipError("x2 must be B x C2 x L")
        C2 = x2.shape[1]
    if wp.dtype != torch.int8:
        raise SonetHipError("pointmlp_h3_segpool: an h3 pack")
    _chk(ids_sorted, "ids_sorted", torch.int32, 2)
    _chk(pos0, "pos0", torch.int32, 1)
    if tuple(ids_sorted.shape) != (B, L) or pos0.shape[0] != B:
        raise SonetHipError("ids_sorted must be B x L and pos0 B")
    if row_max is not None:
        _chk(row_max, "row_max", torch.int32, 2)
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    dev = _same_device(x1, x2, wp, scale, shift, ids_sorted, pos0, row_max)
    lib = _lib.load()
    if wp.numel() != lib.sonet_pointmlp_x3_pack_size(C1 + C2, Cout):
        raise SonetHipError("packed weight does not match Cin=%d Cout=%d" % (C1 + C2, Cout))
    idx = torch.empty((B, Cout, int(M)), dtype=torch.int32, device=dev)
    val = torch.empty((B, Cout, int(M)), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.sonet_pointmlp_h3_segpool_ws_size(B, Cout, int(M)),), dtype=torch.uint8, device=dev)
    name = "pointmlph3_segpool_%dx%d_L%d" % (C1 + C2, Cout, L)
    _range_arm(name)
    xa = _xaff_args(xaff, C1, C2, dev) if xaff is not None else (None, None, None, None, 0)
    with _lib.on_device(dev), _timed(name + ("_xaff" if xaff is not None else "")):
        check(lib.sonet_pointmlp_h3_segpool_f32(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(ids_sorted), ptr(pos0),
                                                ptr(row_max), int(M), ptr(ws), ptr(idx), ptr(val), B, Cout, L, *xa, stream_ptr()),
              "sonet_pointmlp_h3_segpool_f32")
    return idx, val


def index_max_gather_p16(planes, index, K, row_max=None):
    """index_max_gather on an activation that exists only as P16 planes (``P16``): values (hi + mid) / 32, the 22-bit values the next
    layer multiplies -> (idx i32, val f32) B x C x K."""
    _chk(index, "index", torch.int32, 2)
    B, C, Np = planes.shape
    if tuple(index.shape) != (B, Np):
        raise SonetHipError("index must be B x N' = %s, got %s" % ((B, Np), tuple(index.shape)))
    if row_max is not None:
        _chk(row_max, "row_max", torch.int32, 2)
    dev = _same_device(planes.data, index, row_max)
    idx = torch.empty((B, C, int(K)), dtype=torch.int32, device=dev)
    val = torch.empty((B, C, int(K)), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("index_max_gather_p16"):
        check(_lib.load().sonet_index_max_gather_p16(ptr(planes.data), ptr(index), ptr(row_max), ptr(idx), ptr(val), B, C, Np, int(K), stream_ptr()),
              "sonet_index_max_gather_p16")
    return idx, val


# ------------------------------------------------------------------------------------------ SOM
class SomAssignment:
    """Result of som_assign: node ids per point copy plus the per-node count / sum state."""
    __slots__ = ("B", "N", "M", "k", "min_idx_i32", "min_idx_i64", "count", "sum_ws")


def som_assign(x, node, k, want_i64=False):
    _chk(x, "x", torch.float32, 3)
    _chk(node, "node", torch.float32, 3)
    B, D, N = x.shape
    if D != 3 or node.shape[0] != B or node.shape[1] != 3:
        raise SonetHipError("x must be B x 3 x N and node B x 3 x M, got %s and %s" % (tuple(x.shape), tuple(node.shape)))
    M = node.shape[2]
    dev = _same_device(x, node)
    r = SomAssignment()
    r.B, r.N, r.M, r.k = B, N, M, int(k)
    r.min_idx_i32 = torch.empty((B, r.k * N), dtype=torch.int32, device=dev)
    r.min_idx_i64 = torch.empty((B, r.k * N), dtype=torch.int64, device=dev) if want_i64 else None
    # one allocation, sums first (8-byte aligned), counts right behind: the library clears both with a single memset
    ws = torch.empty((B * 3 * M * 8 + B * M * 4,), dtype=torch.uint8, device=dev)
    r.sum_ws = ws[:B * 3 * M * 8].view(torch.float64).view(B, 3, M)
    r.count = ws[B * 3 * M * 8:].view(torch.int32).view(B, M)
    with _lib.on_device(dev), _timed("som_assign"):
        check(_lib.load().sonet_som_assign_f32(ptr(x), ptr(node), B, N, M, r.k, ptr(r.min_idx_i32), ptr(r.min_idx_i64),
                                               ptr(r.count), ptr(r.sum_ws), stream_ptr()), "sonet_som_assign_f32")
    return r


def som_group(x, sn, a, want_centers=False, want_decentered=False, want_augmented=False):
    """-> dict(som_node Bx3xM, row_max BxM i32, [centers], [x_decentered] Bx3xkN, [x_augmented] Bx6xkN)."""
    _chk(x, "x", torch.float32, 3)
    if sn is not None:
        _chk(sn, "sn", torch.float32, 3)
        if sn.shape != x.shape:
            raise SonetHipError("sn must have the shape of x")
    dev = _same_device(x, sn, a.min_idx_i32)
    B, N, M, k = a.B, a.N, a.M, a.k
    kN = k * N
    out = dict(som_node=torch.empty((B, 3, M), dtype=torch.float32, device=dev),
               row_max=torch.empty((B, M), dtype=torch.int32, device=dev))
    out["centers"] = torch.empty((B, 3, kN), dtype=torch.float32, device=dev) if want_centers else None
    out["x_decentered"] = torch.empty((B, 3, kN), dtype=torch.float32, device=dev) if want_decentered else None
    out["x_augmented"] = torch.empty((B, 6, kN), dtype=torch.float32, device=dev) if want_augmented else None
    with _lib.on_device(dev), _timed("som_group"):
        check(_lib.load().sonet_som_group_f32(ptr(x), ptr(sn), ptr(a.min_idx_i32), ptr(a.count), ptr(a.sum_ws),
                                              B, N, M, k, ptr(out["som_node"]), ptr(out["row_max"]), ptr(out["centers"]),
                                              ptr(out["x_decentered"]), ptr(out["x_augmented"]), stream_ptr()),
              "sonet_som_group_f32")
    return out


def som_sort_group(x, sn, a):
    """Node-sorted grouping for the fused no-grad path -> dict(som_node, row_max, x_aug_sorted Bx6xkN, ids_sorted BxkN, pos0 B)."""
    _chk(x, "x", torch.float32, 3)
    _chk(sn, "sn", torch.float32, 3)
    dev = _same_device(x, sn, a.min_idx_i32)
    B, N, M, k = a.B, a.N, a.M, a.k
    kN = k * N
    out = dict(som_node=torch.empty((B, 3, M), dtype=torch.float32, device=dev),
               row_max=torch.empty((B, M), dtype=torch.int32, device=dev),
               x_aug_sorted=torch.empty((B, 6, kN), dtype=torch.float32, device=dev),
               ids_sorted=torch.empty((B, kN), dtype=torch.int32, device=dev),
               pos0=torch.empty((B,), dtype=torch.int32, device=dev),
               node_off=torch.empty((B, M), dtype=torch.int32, device=dev), count=a.count)
    cursor = torch.empty((B, M), dtype=torch.int32, device=dev)
    with _lib.on_device(dev), _timed("som_sort_group"):
        check(_lib.load().sonet_som_sort_group_f32(ptr(x), ptr(sn), ptr(a.min_idx_i32), ptr(a.count), ptr(a.sum_ws), B, N, M, k,
                                                   ptr(out["som_node"]), ptr(out["row_max"]), ptr(out["x_aug_sorted"]),
                                                   ptr(out["ids_sorted"]), ptr(out["pos0"]), ptr(out["node_off"]), ptr(cursor), stream_ptr()),
              "sonet_som_sort_group_f32")
    return out


def som_assign_sort(x, sn, node, k, want_i64=False, knn=None, deterministic=False):
    """som_assign + som_sort_group of the no-grad pooled path in two launches (``sonet_som_assign_sort_f32``): -> (SomAssignment,
    dict(som_node, row_max, x_aug_sorted, ids_sorted, pos0, node_off, count)).  Node ids / counts bit-identical to the separate calls.
    knn = (knn_I B x M x KI int64, K, center_avg): the second launch also does ``knn_stage_prepare`` on the cluster means
    (``sonet_som_assign_sort_knn_f32``); its result is the dict's "knn_prep".
    deterministic (without knn): the order INSIDE a node does not depend on the arrival order of atomics (``sonet_som_assign_sort_det_f32``):
    the same sorted copy in every run."""
    _chk(x, "x", torch.float32, 3)
    _chk(sn, "sn", torch.float32, 3)
    _chk(node, "node", torch.float32, 3)
    B, D, N = x.shape
    if D != 3 or node.shape[0] != B or node.shape[1] != 3 or sn.shape != x.shape:
        raise SonetHipError("x, sn must be B x 3 x N and node B x 3 x M, got %s, %s and %s" % (tuple(x.shape), tuple(sn.shape), tuple(node.shape)))
    M = node.shape[2]
    dev = _same_device(x, sn, node)
    k = int(k)
    kN = k * N
    lib = _lib.load()
    r = SomAssignment()
    r.B, r.N, r.M, r.k = B, N, M, k
    r.min_idx_i32 = torch.empty((B, kN), dtype=torch.int32, device=dev)
    r.min_idx_i64 = torch.empty((B, kN), dtype=torch.int64, device=dev) if want_i64 else None
    r.sum_ws = torch.empty((B, 3, M), dtype=torch.float64, device=dev)
    r.count = torch.empty((B, M), dtype=torch.int32, device=dev)
    out = dict(som_node=torch.empty((B, 3, M), dtype=torch.float32, device=dev),
               row_max=torch.empty((B, M), dtype=torch.int32, device=dev),
               x_aug_sorted=torch.empty((B, 6, kN), dtype=torch.float32, device=dev),
               ids_sorted=torch.empty((B, kN), dtype=torch.int32, device=dev),
               pos0=torch.empty((B,), dtype=torch.int32, device=dev),
               node_off=torch.empty((B, M), dtype=torch.int32, device=dev), count=r.count)
    ws = torch.empty((lib.sonet_som_assign_sort_ws_size(B, N, M, k),), dtype=torch.uint8, device=dev)
    if knn is not None:
        knn_I, K, avg = knn
        _chk(knn_I, "knn_I", torch.int64, 3)
        KI = knn_I.shape[2]
        if tuple(knn_I.shape[:2]) != (B, M) or KI < K or not 1 <= K <= 128:
            raise SonetHipError("som_assign_sort: knn_I B x M x (>= K), 1 <= K <= 128")
        _same_device(x, knn_I)
        Lm = node_stage_columns(B, M)
        Lp = int(lib.sonet_knn_stage_columns(B, M, int(K)))
        prep = dict(center=torch.empty((B, 3, M), dtype=torch.float32, device=dev), center_p16=p16_flat(3, Lm, B * M, dev),
                    rec=torch.empty((Lp, 4), dtype=torch.int32, device=dev), B=B, M=M, K=int(K), G=min(16, 128 // int(K)), Lp=Lp)
        with _lib.on_device(dev), _timed("som_assign_sort"):
            check(lib.sonet_som_assign_sort_knn_f32(ptr(x), ptr(sn), ptr(node), B, N, M, k, ptr(r.min_idx_i32), ptr(r.min_idx_i64), ptr(r.count),
                                                    ptr(r.sum_ws), ptr(out["som_node"]), ptr(out["row_max"]), ptr(out["x_aug_sorted"]),
                                                    ptr(out["ids_sorted"]), ptr(out["pos0"]), ptr(out["node_off"]), ptr(ws),
                                                    ptr(knn_I), KI, int(K), int(bool(avg)), ptr(prep["center"]), ptr(prep["center_p16"].data), ptr(prep["rec"]),
                                                    stream_ptr()), "sonet_som_assign_sort_knn_f32")
        out["knn_prep"] = prep
        return r, out
    fn = lib.sonet_som_assign_sort_det_f32 if deterministic else lib.sonet_som_assign_sort_f32
    with _lib.on_device(dev), _timed("som_assign_sort_det" if deterministic else "som_assign_sort"):
        check(fn(ptr(x), ptr(sn), ptr(node), B, N, M, k, ptr(r.min_idx_i32), ptr(r.min_idx_i64), ptr(r.count),
                                            ptr(r.sum_ws), ptr(out["som_node"]), ptr(out["row_max"]), ptr(out["x_aug_sorted"]),
                                            ptr(out["ids_sorted"]), ptr(out["pos0"]), ptr(out["node_off"]), ptr(ws), stream_ptr()),
              "sonet_som_assign_sort_f32")
    return r, out


def som_mask(min_idx_i32, M):
    _chk(min_idx_i32, "min_idx", torch.int32, 2)
    dev = _same_device(min_idx_i32)
    B, kN = min_idx_i32.shape
    mask = torch.empty((B, kN, int(M)), dtype=torch.int32, device=dev)
    with _lib.on_device(dev), _timed("som_mask"):
        check(_lib.load().sonet_som_mask_i32(ptr(min_idx_i32), ptr(mask), B, kN, int(M), stream_ptr()), "sonet_som_mask_i32")
    return mask


def node_gather(feat, min_idx_i32):
    """feat B x C x M f32, min_idx B x kN i32 -> B x C x kN (models/segmenter.py:90-98)."""
    _chk(feat, "feat", torch.float32, 3)
    _chk(min_idx_i32, "min_idx", torch.int32, 2)
    dev = _same_device(feat, min_idx_i32)
    B, C, M = feat.shape
    kN = min_idx_i32.shape[1]
    out = torch.empty((B, C, kN), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("node_gather"):
        check(_lib.load().sonet_node_gather_f32(ptr(feat), ptr(min_idx_i32), ptr(out), B, C, M, kN, stream_ptr()),
              "sonet_node_gather_f32")
    return out


def node_add_affine_act_(t, z, min_idx_i32, scale, shift, relu):
    """t B x C x L <- act((t + z[:, :, min_idx]) * scale + shift) in place; z B x C x M."""
    _chk(t, "t", torch.float32, 3)
    _chk(z, "z", torch.float32, 3)
    _chk(min_idx_i32, "min_idx", torch.int32, 2)
    dev = _same_device(t, z, min_idx_i32, scale, shift)
    B, C, L = t.shape
    if z.shape[0] != B or z.shape[1] != C or min_idx_i32.shape != (B, L):
        raise SonetHipError("z must be B x C x M and min_idx B x L")
    with _lib.on_device(dev), _timed("node_add_affine_act"):
        check(_lib.load().sonet_node_add_affine_act_f32(ptr(t), ptr(z), ptr(min_idx_i32), ptr(scale), ptr(shift), int(bool(relu)),
                                                        B, C, L, z.shape[2], stream_ptr()), "sonet_node_add_affine_act_f32")
    return t


def wgrad_x3(g, x, xaff=None):
    """sum_b g[b] . x[b]^T: g B x Cout x L, x B x Cin x L (f32) -> Cout x Cin f32, bf16 x 3 split of both operands on the matrix cores.
    xaff = (scale, shift, relu): x holds the RAW output of a BatchNorm layer, the operand split applies act(x * scale[c] + shift[c]) first."""
    _chk(g, "g", torch.float32, 3)
    _chk(x, "x", torch.float32, 3)
    dev = _same_device(g, x)
    B, Cout, L = g.shape
    if x.shape[0] != B or x.shape[2] != L:
        raise SonetHipError("wgrad_x3: g B x Cout x L and x B x Cin x L")
    Cin = x.shape[1]
    lib = _lib.load()
    dw = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
    if g.numel() == 0 or x.numel() == 0:
        return dw.zero_()
    ws = torch.empty((lib.sonet_wgrad_x3_ws_size(B, Cout, Cin, L),), dtype=torch.uint8, device=dev)
    if xaff is not None:
        xs, xh, xr = xaff
        _chk(xs, "xaff scale", torch.float32, 1)
        _chk(xh, "xaff shift", torch.float32, 1)
        if xs.numel() != Cin or xh.numel() != Cin:
            raise SonetHipError("wgrad_x3: xaff needs Cin = %d coefficients" % Cin)
        _same_device(g, xs, xh)
        with _lib.on_device(dev), _timed("wgradx3_xaff_%dx%d_L%d" % (Cout, Cin, L)):
            check(lib.sonet_wgrad_x3_xaff_f32(ptr(g), ptr(x), ptr(dw), ptr(ws), B, Cout, Cin, L, ptr(xs), ptr(xh), int(bool(xr)), stream_ptr()),
                  "sonet_wgrad_x3_xaff_f32")
        return dw
    with _lib.on_device(dev), _timed("wgradx3_%dx%d_L%d" % (Cout, Cin, L)):
        check(lib.sonet_wgrad_x3_f32(ptr(g), ptr(x), ptr(dw), ptr(ws), B, Cout, Cin, L, stream_ptr()), "sonet_wgrad_x3_f32")
    return dw


def wgrad_bf16_xaff_ok(B, Cout, Cin, L):
    """Shapes ``wgrad_bf16(..., xaff=...)`` takes: the streaming generation of the kernel (16-byte aligned 8-column groups, a long reduction)."""
    return L % 8 == 0 and B * ((L + 63) // 64) >= 2048


def wgrad_bf16(g, x, xaff=None):
    """sum_b g[b] . x[b]^T for bfloat16 operands: g B x Cout x L, x B x Cin x L -> Cout x Cin f32 (one bf16 MFMA per product, f32
    accumulation and partials, fixed-order reduction: ``sonet_wgrad_bf16``).  xaff = (scale, shift, relu): x holds the RAW output of a
    BatchNorm layer, normalised by the operand path (``sonet_wgrad_bf16_xaff``; ``wgrad_bf16_xaff_ok`` shapes)."""
    _chk(g, "g", torch.bfloat16, 3)
    _chk(x, "x", torch.bfloat16, 3)
    dev = _same_device(g, x)
    B, Cout, L = g.shape
    if x.shape[0] != B or x.shape[2] != L:
        raise SonetHipError("wgrad_bf16: g B x Cout x L and x B x Cin x L")
    Cin = x.shape[1]
    lib = _lib.load()
    dw = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
    if g.numel() == 0 or x.numel() == 0:
        return dw.zero_()
    ws = torch.empty((lib.sonet_wgrad_bf16_ws_size(B, Cout, Cin, L),), dtype=torch.uint8, device=dev)
    if xaff is not None:
        xs, xh, xr = xaff
        _chk(xs, "xaff scale", torch.float32, 1)
        _chk(xh, "xaff shift", torch.float32, 1)
        if xs.numel() != Cin or xh.numel() != Cin:
            raise SonetHipError("wgrad_bf16: xaff needs Cin = %d coefficients" % Cin)
        _same_device(x, xs, xh)
        with _lib.on_device(dev), _timed("wgradbf16_%dx%d_L%d_xaff" % (Cout, Cin, L)):
            check(lib.sonet_wgrad_bf16_xaff(ptr(g), ptr(x), ptr(dw), ptr(ws), B, Cout, Cin, L, ptr(xs), ptr(xh), int(bool(xr)), stream_ptr()),
                  "sonet_wgrad_bf16_xaff")
        return dw
    with _lib.on_device(dev), _timed("wgradbf16_%dx%d_L%d" % (Cout, Cin, L)):
        check(lib.sonet_wgrad_bf16(ptr(g), ptr(x), ptr(dw), ptr(ws), B, Cout, Cin, L, stream_ptr()), "sonet_wgrad_bf16")
    return dw


def node_gather_lead_affine_act(z, gidx, lead, wl, scale, shift, relu):
    """act((z[:, :, gidx] + wl . lead) * scale + shift): z B x C x M (the layer on the M node features), gidx B x L i32 (out of
    range -> 0), lead B x NL x L (NL <= 4 per-column channels), wl C x NL.  -> B x C x L f32."""
    _chk(z, "z", dim=3)
    if z.dtype not in (torch.float32, torch.bfloat16):
        raise SonetHipError("node_gather_lead_affine_act: z float32 or bfloat16")
    _chk(gidx, "gidx", torch.int32, 2)
    _chk(lead, "lead", torch.float32, 3)
    _chk(wl, "wl", torch.float32, 2)
    dev = _same_device(z, gidx, lead, wl, scale, shift)
    B, C, M = z.shape
    L, NL = gidx.shape[1], lead.shape[1]
    if gidx.shape[0] != B or lead.shape[0] != B or lead.shape[2] != L or tuple(wl.shape) != (C, NL) or scale.numel() != C or shift.numel() != C:
        raise SonetHipError("node_gather_lead_affine_act: z B x C x M, gidx B x L, lead B x NL x L, wl C x NL, scale / shift C")
    out = torch.empty((B, C, L), dtype=z.dtype, device=dev)
    lib = _lib.load()
    fn = lib.sonet_node_gather_lead_affine_act_f32 if z.dtype == torch.float32 else lib.sonet_node_gather_lead_affine_act_bf16
    with _lib.on_device(dev), _timed("node_gather_lead%s_%dx%d_L%d" % ("" if z.dtype == torch.float32 else "bf16", NL, C, L)):
        check(fn(ptr(z), ptr(gidx), ptr(lead), ptr(wl), ptr(scale), ptr(shift), int(bool(relu)), ptr(out), B, C, L, M, NL, stream_ptr()),
              "sonet_node_gather_lead_affine_act")
    return out


def knn_self(node, K):
    """node B x 3 x M f32 -> B x M x K i64: the K nearest nodes of every node (itself first)."""
    _chk(node, "node", torch.float32, 3)
    dev = _same_device(node)
    B, _, M = node.shape
    out = torch.empty((B, M, int(K)), dtype=torch.int64, device=dev)
    with _lib.on_device(dev), _timed("knn_self"):
        check(_lib.load().sonet_knn_self_f32(ptr(node), ptr(out), B, M, int(K), stream_ptr()), "sonet_knn_self_f32")
    return out


def knn_group(coord, feat, knn_I, center_avg):
    """coord B x 3 x M, feat B x C x M, knn_I B x M x K i64 -> (center B x 3 x M, out B x (3+C) x M x K)."""
    _chk(coord, "coord", torch.float32, 3)
    _chk(feat, "feat", torch.float32, 3)
    _chk(knn_I, "knn_I", torch.int64, 3)
    dev = _same_device(coord, feat, knn_I)
    B, C, M = feat.shape
    K = knn_I.shape[2]
    center = torch.empty((B, 3, M), dtype=torch.float32, device=dev)
    out = torch.empty((B, 3 + C, M, K), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("knn_group"):
        check(_lib.load().sonet_knn_group_f32(ptr(coord), ptr(feat), ptr(knn_I), B, C, M, K, int(bool(center_avg)), ptr(center), ptr(out),
                                              stream_ptr()), "sonet_knn_group_f32")
    return center, out


def lastdim_max(x):
    """max over the last (contiguous) axis, values only; NaN propagates like torch.amax."""
    if x.dtype == torch.bfloat16:                     # node-level tensors (B x C x M): widening is exact and cheap
        x = x.float()
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise SonetHipError("lastdim_max needs a contiguous float32 tensor")
    dev = _same_device(x)
    K = x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    with _lib.on_device(dev), _timed("lastdim_max"):
        check(_lib.load().sonet_lastdim_max_f32(ptr(x), ptr(out), out.numel(), K, stream_ptr()), "sonet_lastdim_max_f32")
    return out


def knn_gather(x, knn_I):
    """x BxCxM f32, knn_I BxMxK i64 -> BxCxMxK (models/operations.py:38-54)."""
    _chk(x, "som_node", torch.float32, 3)
    _chk(knn_I, "som_node_knn_I", torch.int64, 3)
    B, C, M = x.shape
    if knn_I.shape[0] != B or knn_I.shape[1] != M:
        raise SonetHipError("knn_I must be B x M x K, got %s for x %s" % (tuple(knn_I.shape), tuple(x.shape)))
    K = knn_I.shape[2]
    dev = _same_device(x, knn_I)
    out = torch.empty((B, C, M, K), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("knn_gather"):
        check(_lib.load().sonet_knn_gather_f32(ptr(x), ptr(knn_I), ptr(out), B, C, M, K, stream_ptr()), "sonet_knn_gather_f32")
    return out


def knn_gather_bwd(g, knn_I, M):
    """Backward of knn_gather: g B x C x M' x K (f32 / bf16), knn_I B x M' x K i64 -> B x C x M f32 (fixed summation order)."""
    if g.dtype not in (torch.float32, torch.bfloat16):
        raise SonetHipError("knn_gather_bwd: f32 or bf16 gradient")
    _chk(g, "g", g.dtype, 4)
    _chk(knn_I, "som_node_knn_I", torch.int64, 3)
    B, C, Mq, K = g.shape
    if tuple(knn_I.shape) != (B, Mq, K) or Mq != M:
        raise SonetHipError("knn_gather_bwd: knn_I must be B x M x K with the M of the gathered tensor")
    dev = _same_device(g, knn_I)
    lib = _lib.load()
    gx = torch.empty((B, C, M), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.sonet_knn_gather_bwd_ws_size(B, M, K),), dtype=torch.uint8, device=dev)
    fn = lib.sonet_knn_gather_bwd_bf16 if g.dtype == torch.bfloat16 else lib.sonet_knn_gather_bwd_f32
    with _lib.on_device(dev), _timed("knn_gather_bwd"):
        check(fn(ptr(g), ptr(knn_I), ptr(gx), ptr(ws), B, C, M, K, stream_ptr()), "sonet_knn_gather_bwd")
    return gx


class _LastDimMax(torch.autograd.Function):
    """torch.max(x, dim=-1).values with its single-arg-max routing: forward = one kernel (value + index of the first maximum),
    backward = one kernel that writes the whole gradient (no fill + scatter)."""

    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        K = xc.shape[-1]
        rows = xc.numel() // K
        out = torch.empty(xc.shape[:-1], dtype=xc.dtype, device=xc.device)
        idx = torch.empty(xc.shape[:-1], dtype=torch.int32, device=xc.device)
        lib = _lib.load()
        fn = lib.sonet_lastdim_argmax_bf16 if xc.dtype == torch.bfloat16 else lib.sonet_lastdim_argmax_f32
        with _lib.on_device(xc.device), _timed("lastdim_argmax"):
            check(fn(ptr(xc), ptr(out), ptr(idx), rows, K, stream_ptr()), "sonet_lastdim_argmax")
        ctx.save_for_backward(idx)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gc = g.contiguous()
        gx = torch.empty(tuple(gc.shape) + (ctx.K,), dtype=gc.dtype, device=gc.device)
        lib = _lib.load()
        fn = lib.sonet_lastdim_max_bwd_bf16 if gc.dtype == torch.bfloat16 else lib.sonet_lastdim_max_bwd_f32
        with _lib.on_device(gc.device), _timed("lastdim_max_bwd"):
            check(fn(ptr(gc), ptr(idx), ptr(gx), gc.numel(), ctx.K, stream_ptr()), "sonet_lastdim_max_bwd")
        return gx


def lastdim_max_autograd(x):
    """max over the last dimension of a CUDA f32 / bf16 tensor, differentiable (gradient to the FIRST maximum, as torch.max)."""
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or x.numel() == 0:
        return torch.max(x, dim=-1)[0]
    return _LastDimMax.apply(x)


# ------------------------------------------------------------------------------------------ pointmlp
import os as _os

# arithmetic of the fused point-wise layer:
#   "f32": exact-f32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an f32 fma chain
#   "x3" : bf16 MFMA on a 3-way bf16 split of both operands (six terms), f32 accumulate: f32-class accuracy (the
#          reference fixtures are met at the same 1e-5 tolerance), f32 operand range, 1.4-1.7x faster than "f32"
#   "bf16": bf16 STORAGE of the activations and one bf16 MFMA per product, f32 accumulate (BASELINE configs[1]); reduced
#          precision by design (features within ~1e-2 of the f32 reference), indices bit-exact on the same bf16 data
#   "h3" : fp16 MFMA on a two-piece fp16 split with scaled residuals (three terms): the same accuracy at half the
#          matrix work, fp16 operand RANGE (|x| <= 2047 clamped, relative precision fades below ~1e-4) -- DEFAULT for
#          the forward layers (coordinates, normalised activations); gradients (dgrad) always use "x3"
POINTMLP_PRECISION = _os.environ.get("SONET_POINTMLP_PRECISION", "h3")


# run the encoder's first PointNet (eval mode, "x3" arithmetic) as one fused kernel
FUSE_POINTRESNET = _os.environ.get("SONET_FUSE_POINTRESNET", "1") != "0"
# ... and pool it per node in the same kernel (no first_pn_out in HBM unless a caller reads the attribute)
FUSE_POOL = _os.environ.get("SONET_FUSE_POOL", "1") != "0"
# no-grad node-level stage (KNNModule + final PointNet) without the gathered tensor and without concats: the neighbour
# gather happens in the first layer's operand loads, narrow leading panels go last through a rotated weight pack (h3 only)
GATHER_NODE_STAGE = _os.environ.get("SONET_GATHER_NODE_STAGE", "1") != "0"
# KNNModule layer 1 as (layer on the M node features) + gather + coordinate channels instead of a layer over K * M gathered columns
NODE_LINEAR_SPLIT = _os.environ.get("SONET_NODE_LINEAR_SPLIT", "1") != "0"
# no-grad h3 chains of point-wise layers hand their activations on pre-split (P16 planes, csrc/pointmlp_h3p.hip) instead of as f32
P16_CHAINS = _os.environ.get("SONET_P16_CHAINS", "1") != "0"
# no-grad node-level stage (KNNModule + final PointNet + global max) on the third-generation layer, flat column axis, max-over-group
# epilogues (csrc/node_stage.hip): 5 launches instead of 8.  0 = the round-4 stage (second-generation layers + planes_max / lastdim_max)
NODE_STAGE_P16 = _os.environ.get("SONET_NODE_STAGE_P16", "1") != "0"
# bf16 training: the last layer of the first PointNet + the per-node arg-max pool in one launch, first_pn_out never written (0 = store + index_max)
POOLED_TRAIN_EPILOGUE = _os.environ.get("SONET_POOLED_TRAIN_EPILOGUE", "1") != "0"
# f32-class training: the same on node-sorted columns (sonet_pointmlp_h3_segpool_f32; 0 = store + index_max)
H3_SEGPOOL = _os.environ.get("SONET_H3_SEGPOOL", "1") != "0"
# ... and the hidden layers of the first PointNet hand their RAW outputs on: normalise + ReLU is applied by the consumers' operand loads
# (next layer, weight gradient, pooled weight gradient); the normalised activations are never written (0 = a normalise pass per layer)
H3_NORM_ON_LOAD = _os.environ.get("SONET_H3_NORM_ON_LOAD", "1") != "0"
# the sorted training path takes its assignment + node-sorted grouping from the two-launch SOM stage of the no-grad path (0 = som_assign + som_sort_group)
TRAIN_ASSIGN_SORT = _os.environ.get("SONET_TRAIN_ASSIGN_SORT", "1") != "0"
WGRAD_KERNEL = _os.environ.get("SONET_WGRAD_KERNEL", "1") != "0"       # 0: torch.bmm (hipBLASLt f32) for the dense weight gradients


# ---- operand-range guard of the fp16-split ("h3") arithmetic ------------------------------------------------------------------
# h3 has an fp16 operand RANGE (include/sonet_hip.h, "Range log"): every h3 launch inside a ``range_scope`` reports max |x|,
# max |w| (and, for the fused first PointNet, the largest hidden activation) into its own 8-word slot of a per-device log;
# ``violations()`` reads the log back (ONE 1 KiB copy, synchronising) and names the launches whose results are not f32-class.
# Encoder.forward (models/networks.py) wraps itself in a scope and recomputes the batch in the range-safe "x3" arithmetic when
# the log says so -- so the default arithmetic can never hand back clamped features silently.  SONET_RANGE_GUARD=0 turns the
# whole mechanism off (no log, no read-back).
RANGE_GUARD = _os.environ.get("SONET_RANGE_GUARD", "1") != "0"
_RANGE_SLOTS = 32
_B_2047, _B_65504, _B_XLOW, _B_WLOW, _B_WLOW32 = 0x44FFE000, 0x477FE000, 0x3C800000, 0x3B800000, 0x3E000000   # bits of 2047, 65504, 2^-6, 2^-8, 2^-3
_range_logs = {}            # device index -> int32[_RANGE_SLOTS * 8]
_range_active = None        # the innermost open scope
_range_ptr_set = False      # whether the library currently holds a non-NULL slot pointer for this thread


def _bits_to_float(u):
    import struct
    return struct.unpack("<f", struct.pack("<I", int(u) & 0xFFFFFFFF))[0]


class range_scope:
    """``with ops.range_scope(device) as rs: <h3 launches>`` then ``rs.violations()`` (synchronises; [] = all launches in range).
    Eager scopes of one device share its log: read a scope's result before opening the next one on that device;
    ``private=True`` gives the scope a log of its own (sonet_hip.graph.GraphedForward)."""

    def __init__(self, device, private=False):
        self.device = torch.device(device)
        self.names = []
        self.enabled = RANGE_GUARD and self.device.type == "cuda"
        self.log = None
        self._prev = None
        # private: this scope owns its 1 KiB log (a captured forward bakes the clear and the slot pointers into its HIP graph:
        # graphs replayed concurrently, or an eager scope in between, must not clear or overwrite each other's slots)
        self._own = torch.zeros((_RANGE_SLOTS * 8,), dtype=torch.int32, device=self.device) if (private and self.enabled) else None

    def __enter__(self):
        global _range_active
        self._prev = _range_active
        if self.enabled:
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            log = self._own if self._own is not None else _range_logs.get(idx)
            if log is None:
                log = torch.zeros((_RANGE_SLOTS * 8,), dtype=torch.int32, device=self.device)
                _range_logs[idx] = log
            else:
                log.zero_()                                   # one memset node (captured like any launch under a HIP graph)
            self.log = log
            _range_active = self
        return self

    def __exit__(self, *exc):
        global _range_active
        _range_active = self._prev
        _range_unset()
        return False

    def _slot(self, name):
        i = min(len(self.names), _RANGE_SLOTS - 1)            # more launches than slots: the last slot is shared (max still holds)
        self.names.append(name)
        return self.log.data_ptr() + i * 32

    def violations(self):
        """-> list of (launch name, what) for the launches of this scope whose operands left the fp16-split range."""
        if not self.enabled or not self.names:
            return []
        w = self.log.cpu().view(_RANGE_SLOTS, 8).tolist()     # synchronises with the launches
        shared = {}
        out = []
        for n, name in enumerate(self.names):
            i = min(n, _RANGE_SLOTS - 1)
            if i in shared:
                continue
            shared[i] = True
            x, wt, hid = (v & 0xFFFFFFFF for v in w[i][:3])
            low_ok = n < _RANGE_SLOTS - 1                     # a shared slot cannot tell which launch was small
            if x > _B_2047:
                out.append((name, "max |x| = %g exceeds 2047" % _bits_to_float(x)))
            elif low_ok and 0 < x < _B_XLOW:
                out.append((name, "max |x| = %g is below 2^-6 (fp16 residuals go subnormal)" % _bits_to_float(x)))
            # Weight side.  The second-generation layer keeps fp16(w) (limits 65504 and 2^-8).  The fused first PointNet and the third
            # generation (pointmlph3p*) keep fp16(32 w) + fp16(32 w - hi): |w| <= 2047, and max |w| >= 2^-8 -- below that the residual
            # piece is an fp16 subnormal with absolute error 2^-30 on w, i.e. up to 2^-22 relative (the fused kernel logs 32 |w|).
            if name.startswith("pointresnet_fused"):
                if wt > _B_65504:
                    out.append((name, "max |w| = %g exceeds %g" % (_bits_to_float(wt) / 32.0, 65504.0 / 32.0)))
                elif low_ok and 0 < wt < _B_WLOW32:
                    out.append((name, "max |w| = %g is below 2^-8" % (_bits_to_float(wt) / 32.0)))
            elif name.startswith("pointmlph3p"):
                if wt > _B_2047:
                    out.append((name, "max |w| = %g exceeds 2047" % _bits_to_float(wt)))
                elif low_ok and 0 < wt < _B_WLOW:
                    out.append((name, "max |w| = %g is below 2^-8" % _bits_to_float(wt)))
            elif wt > _B_65504:
                out.append((name, "max |w| = %g exceeds 65504" % _bits_to_float(wt)))
            elif low_ok and 0 < wt < _B_WLOW:
                out.append((name, "max |w| = %g is below 2^-8" % _bits_to_float(wt)))
            if hid > _B_2047:
                out.append((name, "a hidden activation reaches %g > 2047" % _bits_to_float(hid)))
        return out


_range_warned = False
_range_pending = []         # training: (event, pinned host copy, names) of steps whose log has not been looked at yet


def range_warn(bad):
    global _range_warned
    if not _range_warned:
        import warnings
        warnings.warn("sonet_hip: fp16-split (h3) operand range left by %d launch(es), e.g. %s: %s -- recomputing in the "
                      "range-safe x3 arithmetic (SONET_POINTMLP_PRECISION=x3 selects it from the start)"
                      % (len(bad), bad[0][0], bad[0][1]), RuntimeWarning, stacklevel=3)
        _range_warned = True


def _range_defer(scope):
    """Queue a scope's log for a later look: async copy to pinned memory + an event, no host wait."""
    if not scope.enabled or not scope.names or torch.cuda.is_current_stream_capturing():
        return
    host = torch.empty((_RANGE_SLOTS * 8,), dtype=torch.int32, pin_memory=True)
    host.copy_(scope.log, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _range_pending.append((ev, host, list(scope.names)))


range_scope.defer = _range_defer


def range_deferred_check():
    """Look at every queued log whose copy has completed (no waiting).  A violation switches the process to the x3
    arithmetic (with a warning): the steps already taken ran on clamped activations, the following ones do not."""
    global POINTMLP_PRECISION
    while _range_pending and _range_pending[0][0].query():
        _, host, names = _range_pending.pop(0)
        probe = range_scope.__new__(range_scope)
        probe.enabled, probe.names, probe.log = True, names, host
        bad = probe.violations()
        if bad:
            range_warn(bad)
            POINTMLP_PRECISION = "x3"
            _range_pending.clear()
            return bad
    return []


def run_guarded(call, device, can_rerun):
    """Run ``call()`` (a forward made of h3 launches) inside an operand-range scope.

    can_rerun (no-grad forwards): the log is read right away -- one small synchronising copy -- and, if any launch left the
    fp16 range, ``call()`` runs again in the range-safe x3 arithmetic: the caller always gets f32-class results.
    not can_rerun (training forwards, which update BatchNorm statistics and must not run twice): the log is queued and
    looked at, without waiting for the GPU, at the next guarded training call; a violation switches the process to x3.
    Nested calls (an encoder inside a guarded model-level forward, anything inside a HIP-graph capture that opened
    its own scope) join the outer scope."""
    device = torch.device(device)
    if POINTMLP_PRECISION != "h3" or not RANGE_GUARD or device.type != "cuda" or _range_active is not None:
        return call()
    if not can_rerun:
        range_deferred_check()                                           # may switch POINTMLP_PRECISION to "x3"
        if POINTMLP_PRECISION != "h3":
            return call()
        with range_scope(device) as rs:
            out = call()
        rs.defer()
        return out
    with range_scope(device) as rs:
        out = call()
    if torch.cuda.is_current_stream_capturing():
        return out
    bad = rs.violations()
    if bad:
        range_warn(bad)
        with precision("x3"):
            out = call()
    return out


def mark_inference(t):
    """Tag a tensor as the product of an inference call (Encoder.forward(is_train=False) on an eval() encoder): eval-mode
    layers that receive it run their no-autograd kernels even when the caller left autograd enabled."""
    if isinstance(t, torch.Tensor) and not t.requires_grad:
        t._sonet_inference = True
    return t


def is_inference(*ts):
    """True when autograd is off, or every given tensor carries the inference tag and none requires grad."""
    if not torch.is_grad_enabled():
        return True
    return bool(ts) and all(isinstance(t, torch.Tensor) and getattr(t, "_sonet_inference", False) and not t.requires_grad for t in ts)


def _range_unset():
    global _range_ptr_set
    if _range_ptr_set:
        _lib.load().sonet_range_log_set(None)
        _range_ptr_set = False


def _range_arm(name):
    """Right before an h3 launch: point the library at this launch's slot (or at nothing outside a scope)."""
    global _range_ptr_set
    if _range_active is None:
        _range_unset()
        return
    import ctypes
    _lib.load().sonet_range_log_set(ctypes.c_void_p(_range_active._slot(name)))
    _range_ptr_set = True


class precision:
    """``with ops.precision("x3"): ...`` -- temporarily select the point-wise arithmetic (the range guard's fallback)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global POINTMLP_PRECISION
        self.prev = POINTMLP_PRECISION
        POINTMLP_PRECISION = self.mode
        return self

    def __exit__(self, *exc):
        global POINTMLP_PRECISION
        POINTMLP_PRECISION = self.prev
        return False


H3_COLUMN_RATIO = 128.0      # largest allowed max|w[:, c]| / min_c max|w[:, c]| of an fp16-split layer (see h3_weight_ok)
_h3_ratio_warned = False


def h3_weight_ok(weight2d):
    """Per-channel side of the h3 operand-range guard, decided on the WEIGHTS (once per pack, one small reduction).

    The range log checks each launch's max |x| (>= 2^-6) -- per launch, not per channel.  A channel far below the tensor maximum
    keeps its value as fp16(x) + residual, with the scaled residual 32 (x - fp16(x)) stored in fp16: below the fp16 normal range its
    ABSOLUTE error is up to 2^-30, whatever the channel's magnitude.  That only matters for the 1e-5 bound when the layer's weights
    make up for the small channel: the error it feeds into an output is |w_c| 2^-30 against outputs of the order |w_typ| max|x|, i.e.
    relative (|w_c| / |w_typ|) 2^-30 / max|x| <= ratio 2^-24 with the launch guard -- inside 2^-17 (about 1e-5) for ratio <= 2^7.
    So: a weight whose input columns differ by more than H3_COLUMN_RATIO in magnitude is packed for the range-safe x3 arithmetic
    (one warning); everything else cannot meet the adversarial case (tests/test_gpu_round2.py::test_h3_per_channel_range_case)."""
    return bool(h3_weight_ratio_flag(weight2d).item())


def h3_weight_ratio_flag(weight2d):
    """The test of h3_weight_ok as a one-element device tensor (1 = fine), without a host synchronisation: the training path
    (weights change every step) reads it one check late through pinned memory (models/layers.py::_h3_ok)."""
    cm = weight2d.detach().abs().amax(dim=0)
    big = cm.max()
    small = torch.where(cm > 0, cm, big).min()                 # smallest non-zero column (all-zero weight: big == small == 0)
    return (big <= small * H3_COLUMN_RATIO).to(torch.int32).reshape(1)


def h3_ratio_warn(what):
    global _h3_ratio_warned
    if not _h3_ratio_warned:
        import warnings
        warnings.warn("sonet_hip: %s has input columns more than %gx apart in magnitude -- it runs in the range-safe x3 arithmetic "
                      "instead of the fp16 split (sonet_hip.ops.h3_weight_ok)" % (what, H3_COLUMN_RATIO), RuntimeWarning, stacklevel=3)
        _h3_ratio_warned = True


def x3_supported(C1, C2, Cout):
    return Cout % 32 == 0 and (C2 == 0 or C1 % 16 == 0)


def pointmlp_pack(weight2d, mode="f32"):
    """[Cout][Cin] f32 -> packed MFMA A-fragment order (device tensor: float32 for "f32", uint8 bytes for "x3")."""
    _chk(weight2d, "weight", torch.float32, 2)
    dev = _same_device(weight2d)
    Cout, Cin = weight2d.shape
    lib = _lib.load()
    with _lib.on_device(dev):
        if mode == "x3":
            wp = torch.empty((lib.sonet_pointmlp_x3_pack_size(Cin, Cout),), dtype=torch.uint8, device=dev)
            check(lib.sonet_pointmlp_x3_pack(ptr(weight2d), ptr(wp), Cin, Cout, stream_ptr()), "sonet_pointmlp_x3_pack")
        elif mode == "bf16":                                 # bf16 storage / bf16 MFMA (int16 marks the flavour)
            wp = torch.empty((lib.sonet_pointmlp_bf16_pack_size(Cin, Cout) // 2,), dtype=torch.int16, device=dev)
            check(lib.sonet_pointmlp_bf16_pack(ptr(weight2d), ptr(wp), Cin, Cout, stream_ptr()), "sonet_pointmlp_bf16_pack")
        elif mode == "h3":                                   # same size; int8 marks the fp16 flavour
            wp = torch.empty((lib.sonet_pointmlp_x3_pack_size(Cin, Cout),), dtype=torch.int8, device=dev)
            check(lib.sonet_pointmlp_h3_pack(ptr(weight2d), ptr(wp), Cin, Cout, stream_ptr()), "sonet_pointmlp_h3_pack")
        else:
            wp = torch.empty((lib.sonet_pointmlp_pack_size(Cin, Cout),), dtype=torch.float32, device=dev)
            check(lib.sonet_pointmlp_pack_f32(ptr(weight2d), ptr(wp), Cin, Cout, stream_ptr()), "sonet_pointmlp_pack_f32")
    return wp


def pointmlp_pack_transposed(weight2d, lo, Ci, Cp, mode):
    """The pack of W[:, lo:lo + Ci]^T (Ci x Cout, zero rows up to Cp) -- the dgrad's weights -- read from ``weight2d`` (Cout x Cin f32,
    contiguous) in place: no transposed copy, and the pack kernel's lanes run along W's rows.  mode "x3" / "h3" / "bf16"."""
    _chk(weight2d, "weight", torch.float32, 2)
    dev = _same_device(weight2d)
    Cout, Cin = weight2d.shape
    if not (0 <= lo and lo + Ci <= Cin and Ci <= Cp):
        raise SonetHipError("pointmlp_pack_transposed: bad block lo=%d Ci=%d Cp=%d of %d columns" % (lo, Ci, Cp, Cin))
    lib = _lib.load()
    src = weight2d.data_ptr() + 4 * lo
    with _lib.on_device(dev):
        if mode == "bf16":
            wp = torch.empty((lib.sonet_pointmlp_bf16_pack_size(Cout, Cp) // 2,), dtype=torch.int16, device=dev)
            check(lib.sonet_pointmlp_bf16_pack_strided(src, 1, Cin, ptr(wp), Cout, Cp, Ci, stream_ptr()), "sonet_pointmlp_bf16_pack_strided")
        elif mode in ("x3", "h3"):
            wp = torch.empty((lib.sonet_pointmlp_x3_pack_size(Cout, Cp),), dtype=torch.uint8 if mode == "x3" else torch.int8, device=dev)
            fn = lib.sonet_pointmlp_x3_pack_strided if mode == "x3" else lib.sonet_pointmlp_h3_pack_strided
            check(fn(src, 1, Cin, ptr(wp), Cout, Cp, Ci, stream_ptr()), "sonet_pointmlp_x3_pack_strided")
        else:
            raise SonetHipError("pointmlp_pack_transposed: mode %r" % (mode,))
    return wp


# ---- every stale weight pack of a step in one launch -------------------------------------------------------------------------------
PACK_REGISTRY = _os.environ.get("SONET_PACK_REGISTRY", "1") != "0"     # 0: a pack launch per layer and flavour (the round-4 behaviour)


class _PackRegistry:
    """The packed copies of the layers' weights (``pointmlp_pack`` / ``pointmlp_pack_transposed`` results), keyed on (weight storage,
    what is packed).  ``get`` returns the current pack; when the weight has changed since it was built (``_version``: the optimizer bumps
    it once per step) EVERY stale pack of that device is refreshed by one ``sonet_pack_multi`` launch -- the layers' forward packs and
    the transposed packs of their dgrads alike -- into the buffers the entries already own.  Entries die with their weight tensor."""

    def __init__(self):
        self.entries = {}            # key -> dict(ref, version, buf, rec)
        self.tables = {}             # device -> (tuple of keys, device table, total blocks)

    @staticmethod
    def _spec(w2d, what):
        """-> (flavour, byte offset into w2d, rs, cs, Cin_pack, rows, Cout_pack) of ("fwd", mode) or ("t", mode, lo, Ci, Cp)."""
        Cout, Cin = w2d.shape
        fl = {"bf16": 0, "x3": 1, "h3": 2}[what[1]]
        if what[0] == "fwd":
            return fl, 0, Cin, 1, Cin, Cout, Cout
        _, _, lo, Ci, Cp = what
        if not (0 <= lo and lo + Ci <= Cin and Ci <= Cp):
            raise SonetHipError("pack registry: bad block lo=%d Ci=%d Cp=%d of %d columns" % (lo, Ci, Cp, Cin))
        return fl, 4 * lo, 1, Cin, Cout, Ci, Cp                       # W[:, lo:lo + Ci]^T: element (o, c) = W[c][lo + o]

    def get(self, w2d, what):
        """w2d: the layer's weight as a contiguous Cout x Cin f32 CUDA tensor (a view of the parameter: same version counter)."""
        key = (w2d.data_ptr(), w2d.device.index, tuple(w2d.shape), what)
        e = self.entries.get(key)
        if e is not None and e["ref"]() is None:                       # the storage was recycled by another tensor
            e = None
        if e is None:
            e = self._create(key, w2d, what)
        elif e["version"] != w2d._version:
            self.refresh(w2d.device)
        return e["buf"]

    def _create(self, key, w2d, what):
        import weakref
        _chk(w2d, "weight", torch.float32, 2)
        lib = _lib.load()
        fl, off, rs, cs, Cin_p, rows, Cout_p = self._spec(w2d, what)
        dev = w2d.device
        with _lib.on_device(dev):
            if fl == 0:
                buf = torch.empty((lib.sonet_pointmlp_bf16_pack_size(Cin_p, Cout_p) // 2,), dtype=torch.int16, device=dev)
                check(lib.sonet_pointmlp_bf16_pack_strided(w2d.data_ptr() + off, rs, cs, ptr(buf), Cin_p, Cout_p, rows, stream_ptr()), "sonet_pointmlp_bf16_pack_strided")
            else:
                buf = torch.empty((lib.sonet_pointmlp_x3_pack_size(Cin_p, Cout_p),), dtype=torch.uint8 if fl == 1 else torch.int8, device=dev)
                fn = lib.sonet_pointmlp_x3_pack_strided if fl == 1 else lib.sonet_pointmlp_h3_pack_strided
                check(fn(w2d.data_ptr() + off, rs, cs, ptr(buf), Cin_p, Cout_p, rows, stream_ptr()), "sonet_pointmlp_x3_pack_strided")
        KC = int(lib.sonet_pack_multi_kc(fl, Cin_p))
        total = 64 * ((Cout_p + 31) // 32) * KC
        # (the weakref is to the view's base when there is one: views come and go, the parameter stays)
        base = w2d._base if w2d._base is not None else w2d
        e = dict(ref=weakref.ref(base), src=w2d, version=w2d._version, buf=buf,
                 rec=(w2d.data_ptr() + off, buf.data_ptr(), rs, cs, total, Cin_p, rows, KC, fl))
        e["src"] = None                                                 # (no strong reference to the weight: only its address and its weakref)
        self.entries[key] = e
        self.tables.pop(dev.index, None)
        # the entry (and its GPU buffer) goes when the weight goes -- not when some other weight next turns stale (a sweep that builds
        # model after model would otherwise pile up packs of dead weights)
        weakref.finalize(base, self._drop, key, e)
        return e

    def _drop(self, key, e):
        if self.entries.get(key) is e:
            del self.entries[key]
            self.tables.pop(key[1], None)

    def refresh(self, device):
        """One ``sonet_pack_multi`` launch over every stale entry of ``device``."""
        import numpy as np
        di = device.index
        live, dead = [], []
        for key, e in self.entries.items():
            if key[1] != di:
                continue
            t = e["ref"]()
            if t is None:
                dead.append(key)
            elif e["version"] != t._version:
                live.append((key, e, t))
        for key in dead:
            del self.entries[key]
        if dead:
            self.tables.pop(di, None)
        if not live:
            return
        keys = tuple(k for k, _, _ in live)
        tab = self.tables.get(di)
        if tab is None or tab[0] != keys:
            rec = np.zeros((len(live), 9), dtype=np.int64)              # 72 bytes per entry
            blk = 0
            for i, (_, e, _) in enumerate(live):
                src, dst, rs, cs, total, Cin_p, rows, KC, fl = e["rec"]
                nblk = (total + 255) // 256
                rec[i, 0], rec[i, 1], rec[i, 2], rec[i, 3], rec[i, 4] = src, dst, rs, cs, total
                rec[i, 5] = (Cin_p & 0xFFFFFFFF) | (rows << 32)
                rec[i, 6] = (KC & 0xFFFFFFFF) | (fl << 32)
                rec[i, 7] = (blk & 0xFFFFFFFF) | (nblk << 32)
                blk += nblk
            with _lib.on_device(device):
                tab = (keys, torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(device), blk)
            self.tables[di] = tab
        with _lib.on_device(device), _timed("pack_multi_%d" % len(live)):
            check(_lib.load().sonet_pack_multi(ptr(tab[1]), len(live), tab[2], stream_ptr()), "sonet_pack_multi")
        for _, e, t in live:
            e["version"] = t._version


packs = _PackRegistry()


def pointmlp(x1, wp, scale, shift, relu, Cout, x2=None, out=None, gidx=None, acc=None):
    """y = act((W . cat(x1, x2)) * scale + shift); x B x C x L f32.  The kernel follows the packing of ``wp``.
    ``acc`` (bf16 packs, B x Cout x L bf16, even L): another gradient of the same tensor -- y = bf16(float(y) + float(acc)) from the store
    of the launch (``sonet_pointmlp_bf16_acc``: autograd's accumulation without its pass).
    ``gidx`` (B x L i32, h3 packs only): column l of x1 (B x C1 x L1) is taken from x1[:, :, gidx[b, l]] -- zeros when the
    index is out of range -- i.e. the layer runs on the gathered tensor without materialising it."""
    bf16 = wp.dtype == torch.int16
    xdt = torch.bfloat16 if bf16 else torch.float32
    _chk(x1, "x", xdt, 3)
    B, C1, L = x1.shape
    L1 = L
    if gidx is not None:
        _chk(gidx, "gidx", torch.int32, 2)
        if wp.dtype not in (torch.int8, torch.int16) or gidx.shape[0] != B:
            raise SonetHipError("pointmlp: a gather index needs an h3 or bf16 pack and B rows")
        L = gidx.shape[1]
    C2 = 0
    if x2 is not None:
        _chk(x2, "x2", xdt, 3)
        if x2.shape[0] != B or x2.shape[2] != L:
            raise SonetHipError("x2 must be B x C2 x L")
        C2 = x2.shape[1]
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    dev = _same_device(x1, x2, wp, scale, shift, gidx)
    lib = _lib.load()
    h3 = wp.dtype == torch.int8
    x3 = wp.dtype == torch.uint8 or h3
    want = (lib.sonet_pointmlp_bf16_pack_size(C1 + C2, Cout) // 2 if bf16 else lib.sonet_pointmlp_x3_pack_size(C1 + C2, Cout) if x3
            else lib.sonet_pointmlp_pack_size(C1 + C2, Cout))
    if wp.numel() != want:
        raise SonetHipError("packed weight has %d elements, expected %d for Cin=%d Cout=%d" % (wp.numel(), want, C1 + C2, Cout))
    y = out if out is not None else torch.empty((B, Cout, L), dtype=xdt, device=dev)
    if y.numel() == 0:
        return y
    if bf16:
        if y.dtype != torch.bfloat16:
            raise SonetHipError("pointmlp: a bf16 pack writes a bfloat16 output")
        if acc is not None:
            _chk(acc, "acc", torch.bfloat16, 3)
            if tuple(acc.shape) != (B, Cout, L) or gidx is not None or L % 2:
                raise SonetHipError("pointmlp: acc must be B x Cout x L (even L, no gather index)")
            _same_device(x1, acc)
            with _lib.on_device(dev), _timed("pointmlpbf16_acc_%dx%d_L%d" % (C1 + C2, Cout, L)):
                check(lib.sonet_pointmlp_bf16_acc(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(acc), ptr(y),
                                                  B, Cout, L, stream_ptr()), "sonet_pointmlp_bf16_acc")
            return y
        with _lib.on_device(dev), _timed("pointmlpbf16_%dx%d_L%d" % (C1 + C2, Cout, L)):
            if gidx is not None:
                check(lib.sonet_pointmlp_bf16_gather(ptr(x1), C1, L1, ptr(gidx), ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)),
                                                     ptr(y), B, Cout, L, stream_ptr()), "sonet_pointmlp_bf16_gather")
            else:
                check(lib.sonet_pointmlp_bf16(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y),
                                              B, Cout, L, stream_ptr()), "sonet_pointmlp_bf16")
        return y
    fn = lib.sonet_pointmlp_h3_f32 if h3 else lib.sonet_pointmlp_x3_f32 if x3 else lib.sonet_pointmlp_f32
    name = "pointmlp%s_%dx%d_L%d" % ("h3" if h3 else "x3" if x3 else "", C1 + C2, Cout, L)
    if h3:
        _range_arm(name)
    with _lib.on_device(dev), _timed(name):
        if gidx is not None:
            check(lib.sonet_pointmlp_h3_gather_f32(ptr(x1), C1, L1, ptr(gidx), ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)),
                                                   ptr(y), B, Cout, L, stream_ptr()), "sonet_pointmlp_h3_gather_f32")
        else:
            check(fn(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y),
                     B, Cout, L, stream_ptr()), "sonet_pointmlp")
    return y


# f32-class training backward: the BatchNorm / ReLU backward of a layer is applied by the operand load of its input-gradient launch, which also
# writes g_raw for the weight gradient (sonet_pointmlp_x3_bnb_f32; 0 = a pass of its own, sonet_pointwise_bwd_apply_f32)
BNB_ON_LOAD = _os.environ.get("SONET_BNB_ON_LOAD", "1") != "0"


# ... and, when the layer BELOW handed its raw output on (normalise-on-load), the same launch's epilogue can compute that layer's BatchNorm-backward
# sums from its output (no statistics pass over (gy, raw) of the layer below).  OFF: measured slower -- the reduction costs the dgrad launch 0.33-0.44 ms
# for the 0.2-0.3 ms pass it replaces (docs/findings.md R5.9); kept as a tested record.
# VARIANTS build only (the product library compiles neither this epilogue nor the tail below: ``variants_only()``).
BWD_STATS_EPILOGUE = _os.environ.get("SONET_BWD_STATS_EPILOGUE", "0") != "0"


def variants_only():
    """True when the loaded library is the variants build (tools/, tests/variants): the measured-slower records are callable."""
    return hasattr(_lib.load(), "sonet_pooled_dgrad_tail_f32")


# a tensor with two consumers in the first PointNet (the first layer's output): the gradient of the consumer whose backward runs first is added by
# the store of the other's input-gradient launch (models/layers.py ``_GradCarry``; 0 = autograd's accumulation, a pass over three tensors)
GRAD_CARRY = _os.environ.get("SONET_GRAD_CARRY", "1") != "0"


def pointmlp_x3_bnb(gy, raw, wpt, scale, shift, a, b, c0, sc, sh, relu, Cout, want_g_raw=True, below=None, acc=None):
    """(W . g_raw) * scale + shift with g_raw = a * (gy masked by raw * sc + sh > 0 when relu) + b * raw + c0 per input channel, in one pass
    over (gy, raw) -- ``pointwise_bwd_apply`` + ``pointmlp`` on an x3 pack, bit for bit.  -> (y B x Cout x L, g_raw or None[, sums]).
    below = (praw B x Cout x L, psc, psh, prelu): y is gy of the layer below; -> also its BatchNorm-backward sums (float64 [2 Cout], the
    layout of ``pointwise_bwd_stats(..., want_sums=True)``) from the epilogue.
    acc (B x Cout x L, f32): another gradient of the same tensor, computed earlier -- y = the product + acc from the store of this launch (what
    autograd's accumulation of the two would hold, bit for bit); not together with ``below``."""
    _chk(gy, "gy", torch.float32, 3)
    _chk(raw, "raw", torch.float32, 3)
    if raw.shape != gy.shape:
        raise SonetHipError("pointmlp_x3_bnb: gy and raw must have the same shape")
    B, C, L = gy.shape
    if wpt.dtype != torch.uint8:
        raise SonetHipError("pointmlp_x3_bnb: an x3 pack")
    for t in (scale, shift):
        _chk(t, "scale / shift", torch.float32, 1)
    for t in (a, b, c0, sc, sh):
        _chk(t, "coefficient", torch.float32, 1)
        if t.numel() != C:
            raise SonetHipError("pointmlp_x3_bnb: %d coefficients expected" % C)
    dev = _same_device(gy, raw, wpt, scale, shift, a, b, c0, sc, sh)
    lib = _lib.load()
    if wpt.numel() != lib.sonet_pointmlp_x3_pack_size(C, Cout):
        raise SonetHipError("packed weight does not match Cin=%d Cout=%d" % (C, Cout))
    y = torch.empty((B, Cout, L), dtype=torch.float32, device=dev)
    g_raw = torch.empty_like(gy) if want_g_raw else None
    if y.numel() == 0:
        return y, g_raw
    praw = psc = psh = pws = sums = None
    prelu = False
    if acc is not None:
        _chk(acc, "acc", torch.float32, 3)
        if below is not None or tuple(acc.shape) != (B, Cout, L):
            raise SonetHipError("pointmlp_x3_bnb: acc must be B x Cout x L and does not combine with the sums of the layer below")
        _same_device(gy, acc)
        with _lib.on_device(dev), _timed("pointmlpx3_bnba_%dx%d_L%d" % (C, Cout, L)):
            check(lib.sonet_pointmlp_x3_bnb_acc_f32(ptr(gy), ptr(raw), C, ptr(wpt), ptr(scale), ptr(shift), ptr(a), ptr(b), ptr(c0), ptr(sc), ptr(sh),
                                                    int(bool(relu)), ptr(g_raw), ptr(acc), ptr(y), B, Cout, L, stream_ptr()),
                  "sonet_pointmlp_x3_bnb_acc_f32")
        return y, g_raw
    if below is not None:
        praw, psc, psh, prelu = below
        _chk(praw, "praw", torch.float32, 3)
        if tuple(praw.shape) != (B, Cout, L) or psc.numel() != Cout or psh.numel() != Cout:
            raise SonetHipError("pointmlp_x3_bnb: the layer below must be B x Cout x L with Cout coefficients")
        _chk(psc, "psc", torch.float32, 1)
        _chk(psh, "psh", torch.float32, 1)
        _same_device(gy, praw, psc, psh)
        pws = torch.empty((lib.sonet_pointmlp_stats_ws_size(B, Cout, L),), dtype=torch.uint8, device=dev)
        sums = torch.empty((2 * Cout,), dtype=torch.float64, device=dev)
    with _lib.on_device(dev), _timed("pointmlpx3_bnb%s_%dx%d_L%d" % ("s" if below is not None else "", C, Cout, L)):
        check(lib.sonet_pointmlp_x3_bnb_f32(ptr(gy), ptr(raw), C, ptr(wpt), ptr(scale), ptr(shift), ptr(a), ptr(b), ptr(c0), ptr(sc), ptr(sh),
                                            int(bool(relu)), ptr(g_raw), ptr(y), B, Cout, L, ptr(praw), ptr(psc), ptr(psh), int(bool(prelu)),
                                            ptr(pws), ptr(sums), stream_ptr()), "sonet_pointmlp_x3_bnb_f32")
    if below is not None:
        return y, g_raw, sums
    return y, g_raw


def pointmlp_bf16_bnb_ok(C, Cout, L):
    """Shapes ``pointmlp_bf16_bnb`` takes (``sonet_pointmlp_bf16_bnb``)."""
    return C % 16 == 0 and 32 <= C <= 512 and Cout % 64 == 0 and L % 2 == 0


def pointmlp_bf16_bnb(gy, raw, wpt, scale, shift, a, b, c0, sc, sh, relu, Cout, want_g_raw=True, acc=None):
    """bf16((W . g_raw) * scale + shift) [+ acc] with g_raw = bf16(a * (gy masked by raw * sc + sh > 0 when relu) + b * raw + c0) per input
    channel, in one pass over (gy, raw) -- ``pointwise_bwd_apply`` + ``pointmlp`` (or its ``acc`` form) on a bf16 pack, bit for bit
    (``sonet_pointmlp_bf16_bnb``).  -> (y B x Cout x L bf16, g_raw or None)."""
    _chk(gy, "gy", torch.bfloat16, 3)
    _chk(raw, "raw", torch.bfloat16, 3)
    if raw.shape != gy.shape:
        raise SonetHipError("pointmlp_bf16_bnb: gy and raw must have the same shape")
    B, C, L = gy.shape
    if wpt.dtype != torch.int16:
        raise SonetHipError("pointmlp_bf16_bnb: a bf16 pack")
    for t in (scale, shift):
        _chk(t, "scale / shift", torch.float32, 1)
        if t.numel() != Cout:
            raise SonetHipError("pointmlp_bf16_bnb: %d output coefficients expected" % Cout)
    for t in (a, b, c0, sc, sh):
        _chk(t, "coefficient", torch.float32, 1)
        if t.numel() != C:
            raise SonetHipError("pointmlp_bf16_bnb: %d coefficients expected" % C)
    dev = _same_device(gy, raw, wpt, scale, shift, a, b, c0, sc, sh)
    lib = _lib.load()
    if wpt.numel() * wpt.element_size() != lib.sonet_pointmlp_bf16_pack_size(C, Cout):
        raise SonetHipError("packed weight does not match Cin=%d Cout=%d" % (C, Cout))
    if not pointmlp_bf16_bnb_ok(C, Cout, L):
        raise SonetHipError("pointmlp_bf16_bnb: needs C %% 16 == 0, 32 <= C <= 512, Cout %% 64 == 0, even L (got C=%d Cout=%d L=%d)" % (C, Cout, L))
    if acc is not None:
        _chk(acc, "acc", torch.bfloat16, 3)
        if tuple(acc.shape) != (B, Cout, L):
            raise SonetHipError("pointmlp_bf16_bnb: acc must be B x Cout x L")
        _same_device(gy, acc)
    y = torch.empty((B, Cout, L), dtype=torch.bfloat16, device=dev)
    g_raw = torch.empty_like(gy) if want_g_raw else None
    if y.numel() == 0:
        return y, g_raw
    with _lib.on_device(dev), _timed("pointmlpbf16_bnb%s_%dx%d_L%d" % ("a" if acc is not None else "", C, Cout, L)):
        check(lib.sonet_pointmlp_bf16_bnb(ptr(gy), ptr(raw), C, ptr(wpt), ptr(scale), ptr(shift), ptr(a), ptr(b), ptr(c0), ptr(sc), ptr(sh),
                                          int(bool(relu)), ptr(g_raw), ptr(acc), ptr(y), B, Cout, L, stream_ptr()), "sonet_pointmlp_bf16_bnb")
    return y, g_raw


def pointmlp_nodeadd(x1, wp, scale, shift, relu, Cout, z, zidx, x2=None):
    """act((W . cat(x1, x2) + z[:, :, zidx]) * scale + shift) in one launch: z B x Cout x M f32 (the per-node block of the layer's
    pre-activation), zidx B x L i32 (out of range: + 0).  h3 packs only."""
    if wp.dtype != torch.int8:
        raise SonetHipError("pointmlp_nodeadd: an h3 pack")
    _chk(x1, "x", torch.float32, 3)
    _chk(z, "z", torch.float32, 3)
    _chk(zidx, "zidx", torch.int32, 2)
    B, C1, L = x1.shape
    C2 = 0
    if x2 is not None:
        _chk(x2, "x2", torch.float32, 3)
        if x2.shape[0] != B or x2.shape[2] != L:
            raise SonetHipError("x2 must be B x C2 x L")
        C2 = x2.shape[1]
    dev = _same_device(x1, x2, wp, scale, shift, z, zidx)
    lib = _lib.load()
    if wp.numel() != lib.sonet_pointmlp_x3_pack_size(C1 + C2, Cout) or tuple(z.shape[:2]) != (B, Cout) or tuple(zidx.shape) != (B, L):
        raise SonetHipError("pointmlp_nodeadd: pack for Cin=%d Cout=%d, z B x Cout x M, zidx B x L" % (C1 + C2, Cout))
    y = torch.empty((B, Cout, L), dtype=torch.float32, device=dev)
    name = "pointmlph3_nodeadd_%dx%d_L%d" % (C1 + C2, Cout, L)
    _range_arm(name)
    with _lib.on_device(dev), _timed(name):
        check(lib.sonet_pointmlp_h3_nodeadd_f32(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, Cout, L,
                                                ptr(z), ptr(zidx), z.shape[2], stream_ptr()), "sonet_pointmlp_h3_nodeadd_f32")
    return y


# backward: the weight gradient of a layer does not depend on its input gradient -- the two run on two HIP streams
BWD_SIDE_STREAM = _os.environ.get("SONET_BWD_SIDE_STREAM", "1") != "0"
# ... of the pooled last layer in particular: its sparse weight gradient beside its sparse input gradient (0 = one after the other)
POOLED_SIDE_STREAM = _os.environ.get("SONET_POOLED_SIDE_STREAM", "1") != "0"
_side_streams = {}


class side_stream:
    """``with ops.side_stream(device) as s: <launches>`` -- the launches go to a per-device side stream that first waits for everything
    queued on the current stream; ``s.join()`` (after the block) makes the current stream wait for them.  Tensors ALLOCATED inside the
    block and used afterwards must be passed to ``s.keep(t)`` (record_stream: the caching allocator then knows the consumer stream).
    Disabled (everything on the current stream) when BWD_SIDE_STREAM is off, on CPU, or while a HIP graph is being captured."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.on = (BWD_SIDE_STREAM and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing())
        self.main = self.side = self._ctx = None

    def __enter__(self):
        if self.on:
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self.side = _side_streams.get(idx)
            if self.side is None:
                self.side = _side_streams[idx] = torch.cuda.Stream(device=self.device)
            self.main = torch.cuda.current_stream(self.device)
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
        return False

    def keep(self, t):
        if self.on and t is not None:
            t.record_stream(self.main)
        return t

    def reads(self, *ts):
        """Tensors the launches of the block READ that were allocated on the main stream and may be released before the side stream is
        joined (``join(defer=True)``): the caching allocator must not hand their memory out again before the side stream has passed this point."""
        if self.on:
            for t in ts:
                if t is not None and t.is_cuda:
                    t.record_stream(self.side)

    def join(self, defer=False):
        """The current stream waits for the side stream.  ``defer=True`` (from an autograd backward only, ``ops.DEFER_WGRAD_JOIN``): the wait is
        left to the END of the backward pass (an engine callback) -- or to whoever reads a gradient earlier and calls
        ``ops.join_side_streams()`` first (the gradient all-reducer's bucket hooks do) --, so that the launches behind this point on the
        current stream (the next layer's backward) run beside the side stream's instead of waiting for them.  Everything the side-stream
        launches read must have gone through ``reads()``."""
        if not self.on:
            return
        idx = self.side.device.index
        task = _graph_task_id() if _graph_task_id is not None else -1     # (-1: not inside an autograd pass, or a torch without the query)
        if defer and DEFER_WGRAD_JOIN and task != -1:
            with _join_lock:
                _pending_join[idx] = (self.main, self.side)
                queue = _join_queued[0] != task              # (once per backward pass; a pass that died with an exception never ran its callback)
                _join_queued[0] = task
            if queue:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_join)
            return
        self.main.wait_stream(self.side)
        with _join_lock:
            _pending_join.pop(idx, None)


# weight gradients are not needed before the backward pass is over: their side-stream launches are joined there (0 = at the end of every
# layer's backward, the behaviour up to round 5's first session)
DEFER_WGRAD_JOIN = _os.environ.get("SONET_DEFER_WGRAD_JOIN", "1") != "0"
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_pending_join = {}               # device index -> (main stream, side stream) with launches nobody has waited for yet (under _join_lock)
_join_queued = [-1]             # id of the autograd graph task whose end-of-pass callback is queued


def join_side_streams():
    """Every stream that handed work to a side stream with a deferred join waits for it now (a no-op when nothing is pending).  Call it
    before reading a weight gradient INSIDE a backward pass (gradient hooks); after ``backward()`` has returned it has already happened."""
    with _join_lock:
        pairs = list(_pending_join.values())
        _pending_join.clear()
    for main, side in pairs:
        main.wait_stream(side)


def _end_of_backward_join():
    with _join_lock:
        _join_queued[0] = -1
    join_side_streams()


STATS_EPILOGUE = _os.environ.get("SONET_STATS_EPILOGUE", "1") != "0"   # 0: BatchNorm batch statistics by a separate pass (channel_stats)


# bf16 training (BASELINE configs[1]): the same for the bf16 layers -- the normalise + ReLU pass of the first PointNet's hidden layers (three
# streams over B x C x kN bf16 tensors per step) is gone; the streaming layer kernel, the pooled last layer and the two weight-gradient kernels
# apply act(raw * scale + shift) to their operands (sonet_pointmlp_bf16_stats_xaff / _pool_xaff, sonet_wgrad_bf16_xaff, sonet_pooled_wgrad_xaff_xbf16)
BF16_NORM_ON_LOAD = _os.environ.get("SONET_BF16_NORM_ON_LOAD", "1") != "0"
# ... and the BatchNorm / ReLU backward of the bf16 hidden layers rides on the operand load of the input-gradient launch (sonet_pointmlp_bf16_bnb)
# instead of being a pass over (gy, raw) that writes g_raw for the launch to read back
BF16_BNB_ON_LOAD = _os.environ.get("SONET_BF16_BNB_ON_LOAD", "1") != "0"


def bf16_xaff_ok(B, C1, C2, Cout, L):
    """Shapes the normalise-on-load form of the bf16 layer takes (the streaming kernel: ``sonet_pointmlp_bf16_stats_xaff``)."""
    return ((C1 + C2) % 64 == 0 and (C2 == 0 or C1 % 16 == 0) and Cout % 64 == 0 and L % 2 == 0 and B * ((L + 63) // 64) >= 8192
            and B * L * Cout * 4 >= (32 << 20))


def xaff_ok(C1, C2, Cout):
    """Shapes the normalise-on-load form of the h3 layer takes (``sonet_pointmlp_h3_stats_xaff_f32`` / the xaff arguments of
    ``sonet_pointmlp_h3_segpool_f32``)."""
    return C1 + C2 <= 1024 and Cout % 128 == 0 and (C2 == 0 or C1 % 16 == 0)


def _xaff_args(xaff, C1, C2, dev):
    """xaff = (s1, h1, relu1[, s2, h2, relu2]) -> the C ABI's five arguments."""
    s1, h1, r1 = xaff[:3]
    s2, h2, r2 = (xaff[3:6] if len(xaff) >= 6 else (None, None, False))
    for t, n in ((s1, C1), (h1, C1), (s2, C2), (h2, C2)):
        if t is None:
            continue
        _chk(t, "xaff", torch.float32, 1)
        if t.numel() != n or t.device != dev:
            raise SonetHipError("xaff: %d coefficients on %s expected" % (n, dev))
    if C2 and (s2 is None or h2 is None):
        raise SonetHipError("xaff: a second input needs its coefficients")
    return ptr(s1), ptr(h1), ptr(s2), ptr(h2), int(bool(r1)) | (int(bool(r2)) << 1)


@_consumes_rider
def pointmlp_stats(x1, wp, scale, shift, relu, Cout, x2=None, xaff=None):
    """pointmlp(...) plus the per-channel (mean, biased var) of its output over (B, L), from the kernel's epilogue.  h3 / x3 packs
    (f32 storage) only; -> (y, mean, var).  xaff (h3 packs): the inputs are RAW outputs of BatchNorm layers, normalised by the operand load."""
    h3 = wp.dtype == torch.int8
    bf16 = wp.dtype == torch.int16
    if not (h3 or bf16 or wp.dtype == torch.uint8):
        raise SonetHipError("pointmlp_stats: an h3, x3 or bf16 pack")
    xdt = torch.bfloat16 if bf16 else torch.float32
    _chk(x1, "x", xdt, 3)
    B, C1, L = x1.shape
    C2 = 0
    if x2 is not None:
        _chk(x2, "x2", xdt, 3)
        if x2.shape[0] != B or x2.shape[2] != L:
            raise SonetHipError("x2 must be B x C2 x L")
        C2 = x2.shape[1]
    dev = _same_device(x1, x2, wp, scale, shift)
    lib = _lib.load()
    want = lib.sonet_pointmlp_bf16_pack_size(C1 + C2, Cout) // 2 if bf16 else lib.sonet_pointmlp_x3_pack_size(C1 + C2, Cout)
    if wp.numel() != want:
        raise SonetHipError("packed weight does not match Cin=%d Cout=%d" % (C1 + C2, Cout))
    y = torch.empty((B, Cout, L), dtype=xdt, device=dev)
    mean = torch.empty((Cout,), dtype=torch.float32, device=dev)
    var = torch.empty((Cout,), dtype=torch.float32, device=dev)
    if bf16:
        ws = torch.empty((lib.sonet_pointmlp_bf16_stats_ws_size(B, Cout, L),), dtype=torch.uint8, device=dev)
        try:
            with _lib.on_device(dev), _timed("pointmlpbf16_stats_%dx%d_L%d%s" % (C1 + C2, Cout, L, "_xaff" if xaff is not None else "")):
                if xaff is not None:
                    check(lib.sonet_pointmlp_bf16_stats_xaff(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, Cout, L,
                                                             ptr(ws), ptr(mean), ptr(var), *_xaff_args(xaff, C1, C2, dev), stream_ptr()),
                          "sonet_pointmlp_bf16_stats_xaff")
                else:
                    check(lib.sonet_pointmlp_bf16_stats(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, Cout, L,
                                                        ptr(ws), ptr(mean), ptr(var), stream_ptr()), "sonet_pointmlp_bf16_stats")
        finally:
            _rider_done()
        return y, mean, var
    ws = torch.empty((lib.sonet_pointmlp_stats_ws_size(B, Cout, L),), dtype=torch.uint8, device=dev)
    name = "pointmlp%s_stats_%dx%d_L%d" % ("h3" if h3 else "x3", C1 + C2, Cout, L)
    if h3:
        _range_arm(name)
    fn = lib.sonet_pointmlp_h3_stats_f32 if h3 else lib.sonet_pointmlp_x3_stats_f32
    try:
        with _lib.on_device(dev), _timed(name + ("_xaff" if xaff is not None else "")):
            if xaff is not None:
                if not h3:
                    raise SonetHipError("pointmlp_stats: normalise-on-load needs an h3 pack")
                check(lib.sonet_pointmlp_h3_stats_xaff_f32(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, Cout, L,
                                                           ptr(ws), ptr(mean), ptr(var), *_xaff_args(xaff, C1, C2, dev), stream_ptr()),
                      "sonet_pointmlp_h3_stats_xaff_f32")
            else:
                check(fn(ptr(x1), C1, ptr(x2), C2, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, Cout, L, ptr(ws), ptr(mean), ptr(var),
                         stream_ptr()), "sonet_pointmlp_stats")
    finally:
        _rider_done()
    return y, mean, var


# ---- third generation of the fp16-split layer: pre-split activations ("P16" planes, csrc/pointmlp_h3p.hip) -------------------------
class P16:
    """A B x C x L activation in the P16 layout (include/sonet_hip.h): two fp16 planes per value, in MFMA B-fragment order.
    ``data`` is the raw byte tensor (64 * ceil(C / 16) * L bytes per cloud).  Produced by ``p16_from_f32`` or by ``pointmlp_h3p(...,
    out="p16")``, consumed by ``pointmlp_h3p``; ``float()`` decodes it."""

    __slots__ = ("data", "B", "C", "L")

    def __init__(self, data, B, C, L):
        self.data, self.B, self.C, self.L = data, B, C, L

    @property
    def shape(self):
        return (self.B, self.C, self.L)

    @property
    def device(self):
        return self.data.device

    def float(self):
        return p16_to_f32(self)


def p16_empty(B, C, L, device):
    n = _lib.load().sonet_p16_size(B, C, L)
    return P16(torch.empty((n,), dtype=torch.uint8, device=device), B, C, L)


def p16_from_f32(x, scale=None, shift=None, relu=False):
    """f32 B x C x L -> P16 planes of act(x * scale + shift) (scale / shift per channel, both or none).  Inside a range scope the
    launch logs the largest magnitude it split (word 2 of its slot: the consumers of the planes cannot check it any more)."""
    _chk(x, "x", torch.float32, 3)
    B, C, L = x.shape
    dev = _same_device(x, scale, shift)
    out = p16_empty(B, C, L, dev)
    if x.numel() == 0:
        return out
    if scale is not None:
        _chk(scale, "scale", torch.float32, 1)
        _chk(shift, "shift", torch.float32, 1)
    name = "p16_from_f32_%d_L%d" % (C, L)
    _range_arm(name)
    with _lib.on_device(dev), _timed(name):
        check(_lib.load().sonet_p16_from_f32(ptr(x), ptr(out.data), B, C, L, ptr(scale), ptr(shift), int(bool(relu)), stream_ptr()),
              "sonet_p16_from_f32")
    return out


def p16_to_f32(p):
    x = torch.empty((p.B, p.C, p.L), dtype=torch.float32, device=p.device)
    if x.numel():
        with _lib.on_device(p.device), _timed("p16_to_f32"):
            check(_lib.load().sonet_p16_to_f32(ptr(p.data), ptr(x), p.B, p.C, p.L, stream_ptr()), "sonet_p16_to_f32")
    return x


def pointmlp_h3p_pack(weight2d):
    """[Cout][Cin] f32 -> the h3p pack (int32 tensor marks the flavour): K slots in P16 channel order, fp16(32 w) + residual."""
    _chk(weight2d, "weight", torch.float32, 2)
    dev = _same_device(weight2d)
    Cout, Cin = weight2d.shape
    lib = _lib.load()
    with _lib.on_device(dev):
        wp = torch.empty((lib.sonet_pointmlp_h3p_pack_size(Cin, Cout) // 4,), dtype=torch.int32, device=dev)
        check(lib.sonet_pointmlp_h3p_pack(ptr(weight2d), ptr(wp), Cin, Cout, stream_ptr()), "sonet_pointmlp_h3p_pack")
    return wp


@_consumes_rider
def pointmlp_h3p(x1, wp, scale, shift, relu, Cout, x2=None, out="f32", gidx=None, z=None, zidx=None, stats=False, tag=None):
    """y = act((W . cat(x1, x2) [+ z[:, :, zidx]]) * scale + shift) on P16 inputs (x1, x2: ``P16``).
    out: "f32" -> B x Cout x L f32 tensor, "p16" -> ``P16``, "both" -> (f32, P16).  gidx (B x L i32): column l of x1 is x1[:, :, gidx[b, l]].
    stats=True (out "f32" only): also the per-channel (mean, biased var) of y over (B, L) -> (y, mean, var)."""
    if not isinstance(x1, P16) or (x2 is not None and not isinstance(x2, P16)):
        raise SonetHipError("pointmlp_h3p: P16 inputs (ops.p16_from_f32)")
    if wp.dtype != torch.int32:
        raise SonetHipError("pointmlp_h3p: an h3p pack (ops.pointmlp_h3p_pack)")
    B, C1, L1 = x1.shape
    L = L1
    if gidx is not None:
        _chk(gidx, "gidx", torch.int32, 2)
        if gidx.shape[0] != B:
            raise SonetHipError("pointmlp_h3p: gidx must have B rows")
        L = gidx.shape[1]
    C2 = 0
    if x2 is not None:
        if x2.B != B or x2.L != L:
            raise SonetHipError("x2 must be B x C2 x L")
        C2 = x2.C
        if C1 % 16 != 0:
            raise SonetHipError("pointmlp_h3p: with a second input C1 must be a multiple of 16")
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    if (z is None) != (zidx is None):
        raise SonetHipError("pointmlp_h3p: z and zidx come together")
    ZM = 0
    if z is not None:
        _chk(z, "z", torch.float32, 3)
        _chk(zidx, "zidx", torch.int32, 2)
        if tuple(z.shape[:2]) != (B, Cout) or tuple(zidx.shape) != (B, L):
            raise SonetHipError("pointmlp_h3p: z must be B x Cout x M and zidx B x L")
        ZM = z.shape[2]
    if out not in ("f32", "p16", "both") or (stats and out != "f32"):
        raise SonetHipError("pointmlp_h3p: out is 'f32', 'p16' or 'both' (statistics: 'f32')")
    dev = _same_device(x1.data, x2.data if x2 is not None else None, wp, scale, shift, gidx, z, zidx)
    lib = _lib.load()
    # the pack's K range is the concatenation of the inputs' 16-channel chunks
    cin_pack = (C1 + C2) if C2 else C1
    if wp.numel() * 4 != lib.sonet_pointmlp_h3p_pack_size(cin_pack, Cout):
        raise SonetHipError("packed weight has %d bytes, expected %d for Cin=%d Cout=%d"
                            % (wp.numel() * 4, lib.sonet_pointmlp_h3p_pack_size(cin_pack, Cout), cin_pack, Cout))
    y = torch.empty((B, Cout, L), dtype=torch.float32, device=dev) if out in ("f32", "both") else None
    yp = p16_empty(B, Cout, L, dev) if out in ("p16", "both") else None
    mean = var = ws = None
    if stats:
        mean = torch.empty((Cout,), dtype=torch.float32, device=dev)
        var = torch.empty((Cout,), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.sonet_pointmlp_h3p_stats_ws_size(B, Cout, L),), dtype=torch.uint8, device=dev)
    if B * L * Cout != 0:
        name = "pointmlph3p%s_%dx%d_L%d" % ("_nodeadd" if z is not None else "_stats" if stats else ("_" + tag) if tag else "", C1 + C2, Cout, L)
        _range_arm(name)
        with _lib.on_device(dev), _timed(name):
            check(lib.sonet_pointmlp_h3p(ptr(x1.data), C1, L1, ptr(gidx), ptr(x2.data) if x2 is not None else None, C2, ptr(wp), ptr(scale), ptr(shift),
                                         int(bool(relu)), ptr(y), ptr(yp.data) if yp is not None else None, B, Cout, L, ptr(z), ptr(zidx), ZM,
                                         ptr(ws), ptr(mean), ptr(var), stream_ptr()), "sonet_pointmlp_h3p")
    if stats:
        return y, mean, var
    return y if out == "f32" else yp if out == "p16" else (y, yp)


def node_stage_columns(B, M):
    """Column count Lm of the M-level tensors of the flat node-level stage: B * M rounded up to a multiple of 128."""
    return (B * M + 127) // 128 * 128


def p16_flat(C, L, nvalid, device):
    """P16 planes of a flat 1 x C x L activation whose first ``nvalid`` columns a kernel will write: the pad columns must read as zeros
    (garbage there would reach the range log), so a padded axis starts zero-filled."""
    if nvalid == L:
        return p16_empty(1, C, L, device)
    n = _lib.load().sonet_p16_size(1, C, L)
    return P16(torch.zeros((n,), dtype=torch.uint8, device=device), 1, C, L)


def pointmlp_h3p_gmax(x1, wp, scale, shift, relu, Cout, GK, G, ngout, x2=None, out="p16", Lout=None):
    """The layer followed by a max over groups of GK consecutive columns, one launch (``sonet_pointmlp_h3p_gmax``): x1 (x2) ``P16`` with
    B == 1 and L % 128 == 0, G groups per 128-column block.  out "p16" -> ``P16`` 1 x Cout x Lout (Lout >= ngout columns, default ngout;
    pad columns zero), "f32" -> f32 tensor ngout x Cout."""
    if not isinstance(x1, P16) or (x2 is not None and not isinstance(x2, P16)):
        raise SonetHipError("pointmlp_h3p_gmax: P16 inputs")
    if wp.dtype != torch.int32:
        raise SonetHipError("pointmlp_h3p_gmax: an h3p pack (ops.pointmlp_h3p_pack)")
    if x1.B != 1 or (x2 is not None and (x2.B != 1 or x2.L != x1.L)):
        raise SonetHipError("pointmlp_h3p_gmax: one flat cloud (B == 1), x2 with the columns of x1")
    if out not in ("p16", "f32"):
        raise SonetHipError("pointmlp_h3p_gmax: out is 'p16' or 'f32'")
    C1, L = x1.C, x1.L
    C2 = x2.C if x2 is not None else 0
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    dev = _same_device(x1.data, x2.data if x2 is not None else None, wp, scale, shift)
    lib = _lib.load()
    if wp.numel() * 4 != lib.sonet_pointmlp_h3p_pack_size(C1 + C2, Cout):
        raise SonetHipError("packed weight has %d bytes, expected %d for Cin=%d Cout=%d"
                            % (wp.numel() * 4, lib.sonet_pointmlp_h3p_pack_size(C1 + C2, Cout), C1 + C2, Cout))
    Lout = int(ngout if Lout is None else Lout)
    y = torch.empty((ngout, Cout), dtype=torch.float32, device=dev) if out == "f32" else None
    yp = p16_flat(Cout, Lout, ngout, dev) if out == "p16" else None
    name = "pointmlph3p_gmax%d_%dx%d_L%d" % (GK, C1 + C2, Cout, int(ngout) * int(GK))      # (the columns that count: without the blocks' padding)
    _range_arm(name)
    with _lib.on_device(dev), _timed(name):
        check(lib.sonet_pointmlp_h3p_gmax(ptr(x1.data), C1, ptr(x2.data) if x2 is not None else None, C2, ptr(wp), ptr(scale), ptr(shift),
                                          int(bool(relu)), Cout, L, int(GK), int(G), int(ngout), Lout, ptr(y), ptr(yp.data) if yp is not None else None,
                                          stream_ptr()), "sonet_pointmlp_h3p_gmax")
    return y if out == "f32" else yp


def knn_stage_prepare(coord, knn_I, K, center_avg):
    """Index / coordinate side of KNNModule on the flat column axis (``sonet_knn_stage_prepare_f32``; needs only the node coordinates):
    coord B x 3 x M f32, knn_I B x M x KI int64 (first K columns used) -> dict(center B x 3 x M f32, center_p16 ``P16`` 1 x 3 x Lm,
    rec int32 Lp x 4, B, M, K, G) with G = min(16, 128 // K) nodes per 128-column block of the K-level tensor."""
    _chk(coord, "coord", torch.float32, 3)
    _chk(knn_I, "knn_I", torch.int64, 3)
    B, three, M = coord.shape
    KI = knn_I.shape[2]
    if three != 3 or tuple(knn_I.shape[:2]) != (B, M) or KI < K or not 1 <= K <= 128:
        raise SonetHipError("knn_stage_prepare: coord B x 3 x M, knn_I B x M x (>= K), 1 <= K <= 128")
    dev = _same_device(coord, knn_I)
    lib = _lib.load()
    Lm = node_stage_columns(B, M)
    Lp = int(lib.sonet_knn_stage_columns(B, M, int(K)))
    center = torch.empty((B, 3, M), dtype=torch.float32, device=dev)
    cp = p16_flat(3, Lm, B * M, dev)
    rec = torch.empty((Lp, 4), dtype=torch.int32, device=dev)
    _range_arm("knn_stage_prepare")
    with _lib.on_device(dev), _timed("knn_stage_prepare"):
        check(lib.sonet_knn_stage_prepare_f32(ptr(coord), ptr(knn_I), KI, int(bool(center_avg)), B, M, int(K), ptr(center), ptr(cp.data), ptr(rec),
                                              stream_ptr()), "sonet_knn_stage_prepare_f32")
    return dict(center=center, center_p16=cp, rec=rec, B=B, M=M, K=int(K), G=min(16, 128 // int(K)), Lp=Lp)


def knn_stage_input(prep, z, wl, scale, shift, relu):
    """KNNModule's first layer per neighbour copy (``sonet_knn_stage_input_p16``): prep from ``knn_stage_prepare``, z ``P16`` 1 x C x Lm
    (the layer's feature block applied per node), wl C x 3 -> h1 ``P16`` 1 x C x Lp."""
    if not isinstance(z, P16) or z.B != 1 or z.L != node_stage_columns(prep["B"], prep["M"]):
        raise SonetHipError("knn_stage_input: z is a P16 1 x C x Lm activation")
    _chk(wl, "wl", torch.float32, 2)
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    C = z.C
    if tuple(wl.shape) != (C, 3) or scale.numel() != C or shift.numel() != C:
        raise SonetHipError("knn_stage_input: wl C x 3, scale / shift C")
    dev = _same_device(prep["rec"], z.data, wl, scale, shift)
    h1 = p16_empty(1, C, prep["Lp"], dev)            # (every column is written: pad columns as zeros)
    name = "knn_stage_input_%dx%d" % (C, prep["B"] * prep["M"] * prep["K"])
    _range_arm(name)
    with _lib.on_device(dev), _timed(name):
        check(_lib.load().sonet_knn_stage_input_p16(ptr(prep["rec"]), ptr(z.data), ptr(wl), ptr(scale), ptr(shift), int(bool(relu)),
                                                    prep["B"], prep["M"], prep["K"], C, ptr(h1.data), stream_ptr()), "sonet_knn_stage_input_p16")
    return h1


def p16_flat_to_bcm(p, B, M):
    """``P16`` 1 x C x Lm on the flat column axis of the node-level stage -> f32 B x C x M."""
    if p.B != 1 or p.L != node_stage_columns(B, M):
        raise SonetHipError("p16_flat_to_bcm: a 1 x C x (B M) activation")
    x = torch.empty((B, p.C, M), dtype=torch.float32, device=p.device)
    with _lib.on_device(p.device), _timed("p16_flat_to_bcm"):
        check(_lib.load().sonet_p16_flat_to_bcm_f32(ptr(p.data), ptr(x), B, p.C, M, stream_ptr()), "sonet_p16_flat_to_bcm_f32")
    return x


def pointresnet_pack(w1, w2, w3, w4):
    """Pack the four [Cout][Cin] f32 weights of the first PointNet into the fused kernel's weight stream."""
    for i, w in enumerate((w1, w2, w3, w4)):
        _chk(w, "w%d" % (i + 1), torch.float32, 2)
    if (w1.shape[0], tuple(w2.shape), tuple(w3.shape), tuple(w4.shape)) != (64, (128, 64), (256, 128), (384, 320)) or w1.shape[1] > 16:
        raise SonetHipError("pointresnet_fused supports Cin0<=16 -> 64 -> 128 -> 256 -> [320] -> 384 only")
    dev = _same_device(w1, w2, w3, w4)
    lib = _lib.load()
    ws = torch.empty((lib.sonet_pointresnet_pack_size(),), dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        check(lib.sonet_pointresnet_pack(ptr(w1), ptr(w2), ptr(w3), ptr(w4), w1.shape[1], ptr(ws), stream_ptr()),
              "sonet_pointresnet_pack")
    return ws


def pointresnet_fused(x, wstream, affine, want_p16=False):
    """x B x Cin0 x L f32 -> B x 384 x L f32 (whole first PointNet, eval BN folded into ``affine`` 832 x 2).
    want_p16: -> (y, P16 planes of y), written by the same launch (the operand format of ``pointmlp_h3p``)."""
    _chk(x, "x", torch.float32, 3)
    _chk(affine, "affine", torch.float32, 2)
    if tuple(affine.shape) != (832, 2):
        raise SonetHipError("affine must be 832 x 2 (scale, shift)")
    dev = _same_device(x, wstream, affine)
    B, Cin0, L = x.shape
    only = want_p16 == "only"                   # -> (None, planes): y is never written in f32 (P16.float() decodes the planes)
    y = None if only else torch.empty((B, 384, L), dtype=torch.float32, device=dev)
    yp = p16_empty(B, 384, L, dev) if want_p16 else None
    if B * L == 0:
        return (y, yp) if want_p16 else y
    _range_arm("pointresnet_fused_L%d" % L)
    with _lib.on_device(dev), _timed("pointresnet_fused%s_L%d" % ("_p16only" if only else "_p16" if want_p16 else "", L)):
        if want_p16:
            check(_lib.load().sonet_pointresnet_fused_p16_f32(ptr(x), Cin0, ptr(wstream), ptr(affine), ptr(y), ptr(yp.data), B, L, stream_ptr()),
                  "sonet_pointresnet_fused_p16_f32")
        else:
            check(_lib.load().sonet_pointresnet_fused_f32(ptr(x), Cin0, ptr(wstream), ptr(affine), ptr(y), B, L, stream_ptr()),
                  "sonet_pointresnet_fused_f32")
    return (y, yp) if want_p16 else y


def pointresnet_fused_pool(sg, wstream, affine, M, want_p16=False):
    """First PointNet + per-node max-pool in one pass over node-sorted points (``sg`` = som_sort_group result)
    -> B x 384 x M f32.  want_p16: -> (that, ``P16`` 1 x 384 x Lm): the same map pre-split on the flat column axis of the node-level stage."""
    x_sorted = sg["x_aug_sorted"]
    _chk(x_sorted, "x_sorted", torch.float32, 3)
    _chk(affine, "affine", torch.float32, 2)
    dev = _same_device(x_sorted, wstream, affine, sg["ids_sorted"], sg["pos0"], sg["node_off"], sg["count"])
    B, Cin0, L = x_sorted.shape
    lib = _lib.load()
    ws = torch.empty((lib.sonet_pointresnet_pool_ws_size(B, L, int(M)),), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 384, int(M)), dtype=torch.float32, device=dev)
    outp = p16_flat(384, node_stage_columns(B, int(M)), B * int(M), dev) if want_p16 else None
    _range_arm("pointresnet_fused_pool_L%d" % L)
    with _lib.on_device(dev), _timed("pointresnet_fused_pool_L%d" % L):
        if want_p16:
            check(lib.sonet_pointresnet_fused_pool_p16_f32(ptr(x_sorted), Cin0, ptr(wstream), ptr(affine), ptr(sg["ids_sorted"]), ptr(sg["pos0"]),
                                                           ptr(sg["node_off"]), ptr(sg["count"]), ptr(ws), ptr(out), ptr(outp.data), B, L, int(M),
                                                           stream_ptr()), "sonet_pointresnet_fused_pool_p16_f32")
        else:
            check(lib.sonet_pointresnet_fused_pool_f32(ptr(x_sorted), Cin0, ptr(wstream), ptr(affine), ptr(sg["ids_sorted"]), ptr(sg["pos0"]),
                                                       ptr(sg["node_off"]), ptr(sg["count"]), ptr(ws), ptr(out), B, L, int(M), stream_ptr()),
                  "sonet_pointresnet_fused_pool_f32")
    return (out, outp) if want_p16 else out


def pointresnet_bf16_pack(w1, w2, w3, w4):
    """Weight stream of the fused bf16 first PointNet (``sonet_pointresnet_bf16_pack``)."""
    for i, w in enumerate((w1, w2, w3, w4)):
        _chk(w, "w%d" % (i + 1), torch.float32, 2)
    if (w1.shape[0], tuple(w2.shape), tuple(w3.shape), tuple(w4.shape)) != (64, (128, 64), (256, 128), (384, 320)) or w1.shape[1] > 16:
        raise SonetHipError("pointresnet_bf16 supports Cin0<=16 -> 64 -> 128 -> 256 -> [320] -> 384 only")
    dev = _same_device(w1, w2, w3, w4)
    lib = _lib.load()
    ws = torch.empty((lib.sonet_pointresnet_bf16_pack_size(),), dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        check(lib.sonet_pointresnet_bf16_pack(ptr(w1), ptr(w2), ptr(w3), ptr(w4), w1.shape[1], ptr(ws), stream_ptr()),
              "sonet_pointresnet_bf16_pack")
    return ws


def pointresnet_bf16(x, wstream, affine):
    """x B x Cin0 x L f32 -> B x 384 x L bfloat16: whole first PointNet in bf16 (eval BN folded into ``affine`` 832 x 2)."""
    _chk(x, "x", torch.float32, 3)
    _chk(affine, "affine", torch.float32, 2)
    if tuple(affine.shape) != (832, 2):
        raise SonetHipError("affine must be 832 x 2 (scale, shift)")
    dev = _same_device(x, wstream, affine)
    B, Cin0, L = x.shape
    y = torch.empty((B, 384, L), dtype=torch.bfloat16, device=dev)
    if y.numel() == 0:
        return y
    with _lib.on_device(dev), _timed("pointresnet_bf16_L%d" % L):
        check(_lib.load().sonet_pointresnet_bf16(ptr(x), Cin0, ptr(wstream), ptr(affine), ptr(y), B, L, stream_ptr()),
              "sonet_pointresnet_bf16")
    return y


def pointresnet_bf16_pool(sg, wstream, affine, M):
    """bf16 first PointNet + per-node max-pool in one pass over node-sorted points (``sg`` = som_sort_group result)
    -> B x 384 x M f32 holding bf16-representable values (the maxima of the bf16 features the store variant writes)."""
    x_sorted = sg["x_aug_sorted"]
    _chk(x_sorted, "x_sorted", torch.float32, 3)
    _chk(affine, "affine", torch.float32, 2)
    dev = _same_device(x_sorted, wstream, affine, sg["ids_sorted"], sg["pos0"], sg["node_off"], sg["count"])
    B, Cin0, L = x_sorted.shape
    lib = _lib.load()
    ws = torch.empty((lib.sonet_pointresnet_bf16_pool_ws_size(B, L, int(M)),), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 384, int(M)), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("pointresnet_bf16_pool_L%d" % L):
        check(lib.sonet_pointresnet_bf16_pool(ptr(x_sorted), Cin0, ptr(wstream), ptr(affine), ptr(sg["ids_sorted"]), ptr(sg["pos0"]),
                                              ptr(sg["node_off"]), ptr(sg["count"]), ptr(ws), ptr(out), B, L, int(M), stream_ptr()),
              "sonet_pointresnet_bf16_pool")
    return out


@_consumes_rider
def channel_stats(y):
    """per-channel (mean, biased var) over (B, L) of y B x C x L (f32 or bf16 storage; f64 sums either way)."""
    _chk(y, "y", dim=3)
    if y.dtype not in (torch.float32, torch.bfloat16):
        raise SonetHipError("channel_stats: float32 or bfloat16, got %s" % y.dtype)
    dev = _same_device(y)
    B, C, L = y.shape
    ws = torch.empty((2 * C,), dtype=torch.float64, device=dev)
    mean = torch.empty((C,), dtype=torch.float32, device=dev)
    var = torch.empty((C,), dtype=torch.float32, device=dev)
    lib = _lib.load()
    fn = lib.sonet_channel_stats_f32 if y.dtype == torch.float32 else lib.sonet_channel_stats_bf16
    try:
        with _lib.on_device(dev), _timed("channel_stats" if y.dtype == torch.float32 else "channel_stats_bf16"):
            check(fn(ptr(y), B, C, L, ptr(ws), ptr(mean), ptr(var), stream_ptr()), "sonet_channel_stats")
    finally:
        _rider_done()
    return mean, var


def channel_affine_act_(y, scale, shift, relu):
    _chk(y, "y", torch.float32, 3)
    dev = _same_device(y, scale, shift)
    B, C, L = y.shape
    with _lib.on_device(dev), _timed("channel_affine_act"):
        check(_lib.load().sonet_channel_affine_act_f32(ptr(y), ptr(scale), ptr(shift), int(bool(relu)), B, C, L, stream_ptr()),
              "sonet_channel_affine_act_f32")
    return y


def channel_affine_act(x, scale, shift, relu):
    """y = act(x * scale[c] + shift[c]) out of place, x B x C x L (f32 or bf16 storage)."""
    _chk(x, "x", dim=3)
    dev = _same_device(x, scale, shift)
    B, C, L = x.shape
    y = torch.empty_like(x)
    lib = _lib.load()
    fn = lib.sonet_channel_affine_act_out_f32 if x.dtype == torch.float32 else lib.sonet_channel_affine_act_out_bf16
    with _lib.on_device(dev), _timed("channel_affine_act" if x.dtype == torch.float32 else "channel_affine_act_bf16"):
        check(fn(ptr(x), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, C, L, stream_ptr()), "sonet_channel_affine_act_out")
    return y


def chunk_mean(h, k):
    """Mean over the k copies of a point: h B x C x (k N) f32 -> B x C x N = c * ((h[..., :N] + h[..., N:2N]) + h[..., 2N:]), c = 1/3 or 0.5
    (models/networks.py:331-336, the reference's order of operations)."""
    _chk(h, "h", torch.float32, 3)
    B, C, L = h.shape
    if k not in (1, 2, 3) or L % k:
        raise SonetHipError("chunk_mean: k in {1, 2, 3} and a length divisible by k")
    dev = _same_device(h)
    out = torch.empty((B, C, L // k), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("chunk_mean"):
        check(_lib.load().sonet_chunk_mean_f32(ptr(h), ptr(out), B * C, L // k, int(k), stream_ptr()), "sonet_chunk_mean_f32")
    return out


_CONST = {}


def const_vec(C, value, device):
    """Cached constant [C] f32 vector (ones / zeros for the kernels' scale / shift arguments)."""
    key = (int(C), float(value), str(device))
    t = _CONST.get(key)
    if t is None:
        t = torch.full((int(C),), float(value), dtype=torch.float32, device=device)
        _CONST[key] = t
    return t


def _rider_done():
    """After a statistics-producing call: the rider was consumed by its finalize launch -- or, if the call failed before launching,
    must not linger for an unrelated later launch."""
    if getattr(_tls, "rider_armed", None) is not None:
        _lib.load().sonet_bn_rider_set(None, None, 0.0, 0.0, 0.0, None, None, None, None, None)
        _tls.rider_armed = None


def bn_rider(gamma, beta, eps, running_mean=None, running_var=None, momentum=0.0, unbias=1.0):
    """Arm the "BatchNorm rider" (``sonet_bn_rider_set``) for the NEXT statistics-producing call of this thread (``pointmlp_stats``,
    ``pointmlp_h3p(stats=True)``, ``channel_stats``): its finalize launch also writes (invstd, scale, shift) -- ``bn_fwd_coeffs`` -- and,
    given the running statistics, updates them in place -- ``bn_running_update_`` --, operation for operation.  -> (invstd, scale, shift)
    tensors that call will fill."""
    dev = _same_device(gamma, beta, running_mean, running_var)
    C = gamma.numel()
    out = torch.empty((3, C), dtype=torch.float32, device=dev)
    if running_mean is not None:
        for t, n in ((running_mean, "running_mean"), (running_var, "running_var")):
            _chk(t, n, torch.float32, 1)
    g_, b_ = gamma.detach().contiguous(), beta.detach().contiguous()
    check(_lib.load().sonet_bn_rider_set(ptr(g_), ptr(b_), float(eps), float(momentum), float(unbias),
                                         ptr(running_mean), ptr(running_var), ptr(out[0]), ptr(out[1]), ptr(out[2])), "sonet_bn_rider_set")
    _tls.rider_armed = (g_, b_, out)                   # (keeps the operands alive until the statistics call has consumed the rider)
    if running_mean is not None:
        # written through raw pointers by the coming launch: move the version counters as an in-place aten op would
        torch.autograd.graph.increment_version(running_mean)
        torch.autograd.graph.increment_version(running_var)
    return out[0], out[1], out[2]


def bn_fwd_coeffs(mean, var, gamma, beta, eps):
    """-> (invstd, scale, shift) of training BatchNorm, one launch."""
    dev = _same_device(mean, var, gamma, beta)
    C = mean.numel()
    out = torch.empty((3, C), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        check(_lib.load().sonet_bn_fwd_coeffs_f32(ptr(mean), ptr(var), ptr(gamma.detach().contiguous()), ptr(beta.detach().contiguous()),
                                                  float(eps), C, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "sonet_bn_fwd_coeffs_f32")
    return out[0], out[1], out[2]


def bn_bwd_coeffs(sums, mean, invstd, gamma, n):
    """sums [2C] f64 of pointwise_bwd_stats -> (a, b, c0, g_gamma, g_beta), one launch."""
    dev = _same_device(sums, mean, invstd, gamma)
    C = mean.numel()
    out = torch.empty((5, C), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        check(_lib.load().sonet_bn_bwd_coeffs_f32(ptr(sums), ptr(mean), ptr(invstd), ptr(gamma.detach().contiguous()), float(n), C,
                                                  ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(out[4]), stream_ptr()),
              "sonet_bn_bwd_coeffs_f32")
    return out[0], out[1], out[2], out[3], out[4]


def pointwise_bwd_stats(gy, raw, scale, shift, relu, want_sums=False):
    """-> (s1, s2) float64 [C]: sum gy*mask, sum gy*mask*raw over (b, l); mask = (raw*scale+shift > 0) if relu."""
    _chk(gy, "gy", dim=3)
    _chk(raw, "raw", gy.dtype, 3)
    dev = _same_device(gy, raw, scale, shift)
    B, C, L = gy.shape
    sums = torch.empty((2 * C,), dtype=torch.float64, device=dev)
    lib = _lib.load()
    fn = lib.sonet_pointwise_bwd_stats_f32 if gy.dtype == torch.float32 else lib.sonet_pointwise_bwd_stats_bf16
    with _lib.on_device(dev), _timed("pointwise_bwd_stats" if gy.dtype == torch.float32 else "pointwise_bwd_stats_bf16"):
        check(fn(ptr(gy), ptr(raw), ptr(scale), ptr(shift), int(bool(relu)), B, C, L, ptr(sums), stream_ptr()), "sonet_pointwise_bwd_stats")
    if want_sums:
        return sums
    return sums[:C], sums[C:]


def pointwise_bwd_apply(gy, raw, scale, shift, relu, a, b, c0):
    """g_raw = a[c] * (gy * mask) + b[c] * raw + c0[c]."""
    _chk(gy, "gy", dim=3)
    _chk(raw, "raw", gy.dtype, 3)
    dev = _same_device(gy, raw, scale, shift, a, b, c0)
    B, C, L = gy.shape
    out = torch.empty_like(gy)
    lib = _lib.load()
    fn = lib.sonet_pointwise_bwd_apply_f32 if gy.dtype == torch.float32 else lib.sonet_pointwise_bwd_apply_bf16
    with _lib.on_device(dev), _timed("pointwise_bwd_apply" if gy.dtype == torch.float32 else "pointwise_bwd_apply_bf16"):
        check(fn(ptr(gy), ptr(raw), ptr(scale), ptr(shift), int(bool(relu)), ptr(a), ptr(b), ptr(c0), ptr(out), B, C, L, stream_ptr()),
              "sonet_pointwise_bwd_apply")
    return out


P16_ONLY = _os.environ.get("SONET_P16_ONLY", "1") != "0"        # segmenter: the fused first PointNet writes only the P16 planes of first_pn_out
POOLED_DGRAD_MFMA = _os.environ.get("SONET_POOLED_DGRAD_MFMA", "1") != "0"        # bf16 outputs: the dense-tile product on the matrix cores (sonet_pooled_dgrad_mfma_bf16) when the shape allows


def pooled_dgrad_mfma_ok(C, C1, C2, L):
    ct = (C1 + C2 + 31) // 32
    return POOLED_DGRAD_MFMA and L % 2 == 0 and C % 16 == 0 and C <= 384 and ct % 2 == 0 and ct <= 12


# the sparse input gradient of the pooled layer on node-sorted columns (f32-class training): the column-0 part of the gradient (empty nodes) and
# the BatchNorm-backward sums of the layer that produced x2 can ride on the store of the launch (sonet_pooled_dgrad_tail_f32) instead of being two
# scatter_add launches and a statistics pass over (gy, raw) of that layer.  OFF: measured slower -- 1.24 ms against 1.00 ms for the three
# launches apart (the store phase reads raw in 128-byte pieces from five workgroups per CU; docs/findings.md R5.14); kept as a tested record.
POOLED_DGRAD_TAIL = _os.environ.get("SONET_POOLED_DGRAD_TAIL", "0") != "0"


def pooled_dgrad_tail_ok(C1, C2, out_dtype):
    return POOLED_DGRAD_TAIL and out_dtype == torch.float32 and (C1 + C2) % 4 == 0 and variants_only()


def pooled_dgrad(g_pooled, pos_i32, weight2d, C1, C2, L, out_dtype=torch.float32, wt_pack=None, col0=None, pos0=None, below=None):
    """Sparse W^T . g for a gradient that exists only at the pooled positions: g_pooled, pos B x C x M -> (gx1 B x C1 x L, gx2 B x C2 x L).
    wt_pack (bf16 outputs only): pointmlp_pack(W^T, "bf16") -> the matrix-core kernel (g and W rounded to bf16).
    col0 (B x (C1 + C2), f32) with pos0 (B, i32): gx[b][:, pos0[b]] += col0[b], by the store of the launch (``pooled_dgrad_tail_ok``).
    below = (raw B x C2 x L, sc, sh, relu): gx2 is gy of the BatchNorm layer whose raw output is ``raw`` -> a third result, its
    BatchNorm-backward sums (float64 [2 C2], the layout of ``pointwise_bwd_stats(..., want_sums=True)``), taken from the tiles as they are stored."""
    _chk(g_pooled, "g_pooled", torch.float32, 3)
    _chk(pos_i32, "pos", torch.int32, 3)
    _chk(weight2d, "weight", torch.float32, 2)
    dev = _same_device(g_pooled, pos_i32, weight2d)
    B, C, M = g_pooled.shape
    lib = _lib.load()
    ws = torch.empty((lib.sonet_pooled_dgrad_ws_size(B, C, M, int(L)),), dtype=torch.uint8, device=dev)
    gx1 = torch.empty((B, C1, int(L)), dtype=out_dtype, device=dev)
    gx2 = torch.empty((B, C2, int(L)), dtype=out_dtype, device=dev) if C2 else None
    if col0 is not None or pos0 is not None or below is not None:
        if not variants_only():
            raise SonetHipError("pooled_dgrad: col0 / below ride on sonet_pooled_dgrad_tail_f32, a measured-slower record of the variants build "
                                "(SONET_HIP_LIB=%s)" % _lib.VARIANTS_PATH)
        if out_dtype != torch.float32 or (C1 + C2) % 4 or wt_pack is not None:
            raise SonetHipError("pooled_dgrad: col0 / below need f32 outputs and C1 + C2 a multiple of 4")
        if (col0 is None) != (pos0 is None):
            raise SonetHipError("pooled_dgrad: col0 and pos0 come together")
        if col0 is not None:
            _chk(col0, "col0", torch.float32, 2)
            _chk(pos0, "pos0", torch.int32, 1)
            if tuple(col0.shape) != (B, C1 + C2) or pos0.numel() != B:
                raise SonetHipError("pooled_dgrad: col0 must be B x (C1 + C2), pos0 B")
            _same_device(g_pooled, col0, pos0)
        raw = sc = sh = tws = sums = None
        relu = False
        if below is not None:
            raw, sc, sh, relu = below
            _chk(raw, "below raw", torch.float32, 3)
            _chk(sc, "below sc", torch.float32, 1)
            _chk(sh, "below sh", torch.float32, 1)
            if C2 == 0 or tuple(raw.shape) != (B, C2, int(L)) or sc.numel() != C2 or sh.numel() != C2:
                raise SonetHipError("pooled_dgrad: below = (raw B x C2 x L, sc, sh, relu) with C2 coefficients")
            _same_device(g_pooled, raw, sc, sh)
            tws = torch.empty((lib.sonet_pooled_dgrad_tail_ws_size(B, C2, int(L)),), dtype=torch.uint8, device=dev)
            sums = torch.empty((2 * C2,), dtype=torch.float64, device=dev)
        with _lib.on_device(dev), _timed("pooled_dgrad_tail%s" % ("_sums" if below is not None else "")):
            check(lib.sonet_pooled_dgrad_tail_f32(ptr(g_pooled), ptr(pos_i32), ptr(weight2d), B, C, M, C1, C2, int(L), ptr(ws), ptr(gx1), ptr(gx2),
                                                  ptr(col0), ptr(pos0), ptr(raw), ptr(sc), ptr(sh), int(bool(relu)), ptr(tws), ptr(sums), stream_ptr()),
                  "sonet_pooled_dgrad_tail_f32")
        return (gx1, gx2, sums) if below is not None else (gx1, gx2)
    if wt_pack is not None and out_dtype == torch.bfloat16 and pooled_dgrad_mfma_ok(C, C1, C2, int(L)):
        if wt_pack.dtype != torch.int16 or wt_pack.numel() != lib.sonet_pointmlp_bf16_pack_size(C, (C1 + C2 + 31) // 32 * 32) // 2:
            raise SonetHipError("pooled_dgrad: wt_pack is not the bf16 pack of W^T")
        with _lib.on_device(dev), _timed("pooled_dgrad_mfma"):
            check(lib.sonet_pooled_dgrad_mfma_bf16(ptr(g_pooled), ptr(pos_i32), ptr(wt_pack), B, C, M, C1, C2, int(L), ptr(ws), ptr(gx1), ptr(gx2),
                                                   stream_ptr()), "sonet_pooled_dgrad_mfma_bf16")
        return gx1, gx2
    fn = lib.sonet_pooled_dgrad_f32 if out_dtype == torch.float32 else lib.sonet_pooled_dgrad_obf16
    with _lib.on_device(dev), _timed("pooled_dgrad"):
        check(fn(ptr(g_pooled), ptr(pos_i32), ptr(weight2d), B, C, M, C1, C2, int(L), ptr(ws), ptr(gx1), ptr(gx2), stream_ptr()),
              "sonet_pooled_dgrad")
    return gx1, gx2


def pooled_wgrad(g_pooled_t, pos_i32_t, x, xaff=None):
    """Sparse wgrad of the pooled last layer: g_pooled_t, pos_t B x M x C (TRANSPOSED entries), x B x Ci x L -> g_W C x Ci
    (= sum over clouds and entries of g * x[:, pos]; per-cloud partials summed in a fixed order).
    xaff = (scale, shift, relu) (f32 x): x holds the RAW output of a BatchNorm layer, normalised on its way into the LDS."""
    _chk(g_pooled_t, "g_pooled_t", torch.float32, 3)
    _chk(pos_i32_t, "pos_t", torch.int32, 3)
    _chk(x, "x", dim=3)
    dev = _same_device(g_pooled_t, pos_i32_t, x)
    B, M, C = g_pooled_t.shape
    Ci, L = x.shape[1], x.shape[2]
    part = torch.empty((B, C, Ci), dtype=torch.float32, device=dev)
    lib = _lib.load()
    if xaff is not None:
        xs, xh, xr = xaff
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise SonetHipError("pooled_wgrad: float32 or bfloat16 rows")
        _chk(xs, "xaff scale", torch.float32, 1)
        _chk(xh, "xaff shift", torch.float32, 1)
        if xs.numel() != Ci or xh.numel() != Ci:
            raise SonetHipError("pooled_wgrad: xaff needs Ci = %d coefficients" % Ci)
        _same_device(x, xs, xh)
        fnx = lib.sonet_pooled_wgrad_xaff_f32 if x.dtype == torch.float32 else lib.sonet_pooled_wgrad_xaff_xbf16
        with _lib.on_device(dev), _timed("pooled_wgrad_xaff"):
            check(fnx(ptr(g_pooled_t), ptr(pos_i32_t), ptr(x), B, C, M, Ci, L, ptr(part), ptr(xs), ptr(xh), int(bool(xr)), stream_ptr()),
                  "sonet_pooled_wgrad_xaff")
        return part.sum(0)
    fn = lib.sonet_pooled_wgrad_f32 if x.dtype == torch.float32 else lib.sonet_pooled_wgrad_xbf16
    with _lib.on_device(dev), _timed("pooled_wgrad"):
        check(fn(ptr(g_pooled_t), ptr(pos_i32_t), ptr(x), B, C, M, Ci, L, ptr(part), stream_ptr()), "sonet_pooled_wgrad")
    return part.sum(0)


def linear_act(x, weight, scale, shift, relu):
    """x B x Cin, weight Cout x Cin -> act((x @ weight^T) * scale + shift), B x Cout (exact f32)."""
    _chk(x, "x", torch.float32, 2)
    _chk(weight, "weight", torch.float32, 2)
    dev = _same_device(x, weight, scale, shift)
    B, Cin = x.shape
    Cout = weight.shape[0]
    y = torch.empty((B, Cout), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("linear_act_%dx%d" % (Cin, Cout)):
        check(_lib.load().sonet_linear_act_f32(ptr(x), ptr(weight), ptr(scale), ptr(shift), int(bool(relu)), ptr(y), B, Cin, Cout, stream_ptr()),
              "sonet_linear_act_f32")
    return y


FC_HEAD = _os.environ.get("SONET_FC_HEAD", "1") != "0"          # training: the heads' B x C layers on sonet_fc_* (one forward, two backward launches per layer)


def fc_head_ok(x, weight):
    """The training kernels of the heads' FC layers take this (x B x Cin, weight Cout x Cin)?"""
    return (FC_HEAD and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_cuda
            and 1 <= x.shape[0] <= 128 and x.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0 and x.shape[1] >= 4 and weight.shape[0] >= 4)


def fc_bn_act_fwd(x, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, relu):
    """Linear + BatchNorm1d (batch statistics; gamma None: no normalisation) + ReLU, training forward: -> (y, xhat, invstd); the running
    statistics are updated in place (models/layers.py:123-166)."""
    _chk(x, "x", torch.float32, 2)
    _chk(weight, "weight", torch.float32, 2)
    dev = _same_device(x, weight)
    B, Cin = x.shape
    Cout = weight.shape[0]
    y = torch.empty((B, Cout), dtype=torch.float32, device=dev)
    xhat = torch.empty((B, Cout), dtype=torch.float32, device=dev) if gamma is not None else None
    invstd = torch.empty((Cout,), dtype=torch.float32, device=dev) if gamma is not None else None
    with _lib.on_device(dev), _timed("fc_bn_act_fwd_%dx%d" % (Cin, Cout)):
        check(_lib.load().sonet_fc_bn_act_fwd_f32(ptr(x), ptr(weight), ptr(bias), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                  float(momentum), float(eps), int(bool(relu)), B, Cin, Cout, ptr(y), ptr(xhat), ptr(invstd),
                                                  stream_ptr()), "sonet_fc_bn_act_fwd_f32")
    return y, xhat, invstd


def fc_bn_act_bwd(gy, y, xhat, invstd, gamma, x, relu, want_dw=True):
    """-> (dz, dW, dbias, dgamma, dbeta): the layer's backward up to its own parameters (dgamma / dbeta None without a norm)."""
    _chk(gy, "gy", torch.float32, 2)
    dev = _same_device(gy, x)
    B, Cout = gy.shape
    Cin = x.shape[1]
    dz = torch.empty((B, Cout), dtype=torch.float32, device=dev)
    dW = torch.empty((Cout, Cin), dtype=torch.float32, device=dev) if want_dw else None
    vec = torch.empty((3, Cout), dtype=torch.float32, device=dev)
    has_bn = gamma is not None
    with _lib.on_device(dev), _timed("fc_bn_act_bwd_%dx%d" % (Cin, Cout)):
        check(_lib.load().sonet_fc_bn_act_bwd_f32(ptr(gy), ptr(y), ptr(xhat), ptr(invstd), ptr(gamma), ptr(x), int(bool(relu)), B, Cin, Cout,
                                                  ptr(dz), ptr(dW), ptr(vec[0]), ptr(vec[1]) if has_bn else None, ptr(vec[2]) if has_bn else None,
                                                  stream_ptr()), "sonet_fc_bn_act_bwd_f32")
    return dz, dW, vec[0], (vec[1] if has_bn else None), (vec[2] if has_bn else None)


def fc_dx(dz, weight):
    """dx B x Cin = dz (B x Cout) . weight (Cout x Cin)."""
    _chk(dz, "dz", torch.float32, 2)
    _chk(weight, "weight", torch.float32, 2)
    dev = _same_device(dz, weight)
    B, Cout = dz.shape
    Cin = weight.shape[1]
    dx = torch.empty((B, Cin), dtype=torch.float32, device=dev)
    with _lib.on_device(dev), _timed("fc_dx_%dx%d" % (Cin, Cout)):
        check(_lib.load().sonet_fc_dx_f32(ptr(dz), ptr(weight), B, Cin, Cout, ptr(dx), stream_ptr()), "sonet_fc_dx_f32")
    return dx


def chamfer_nn(q, db):
    """q B x 3 x Nq, db B x 3 x Nd -> B x Nq i32 nearest database index."""
    _chk(q, "q", torch.float32, 3)
    _chk(db, "db", torch.float32, 3)
    dev = _same_device(q, db)
    B, _, Nq = q.shape
    Nd = db.shape[2]
    nn = torch.empty((B, Nq), dtype=torch.int32, device=dev)
    with _lib.on_device(dev), _timed("chamfer_nn"):
        check(_lib.load().sonet_chamfer_nn_f32(ptr(q), ptr(db), ptr(nn), B, Nq, Nd, stream_ptr()), "sonet_chamfer_nn_f32")
    return nn


def mfma_f16_sustained_rate(random_operands=True, iters=4000, device=None):
    """(TFLOP/s, shader GHz) a pure fp16 MFMA loop holds on the whole chip -- the measuring stick beside the nominal
    matrix peak (``sonet_diag_mfma_f16_rate``; with random operands the rate is power-limited, DESIGN.md finding 8)."""
    import ctypes
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    tf, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0)
    with _lib.on_device(dev):
        _lib.require_device(dev)
        check(_lib.load().sonet_diag_mfma_f16_rate(1 if random_operands else 0, int(iters), ctypes.byref(tf), ctypes.byref(ghz), stream_ptr()),
              "sonet_diag_mfma_f16_rate")
    return tf.value, ghz.value


def knn_prepare(coord, knn_I, center_avg):
    """coord B x 3 x M, knn_I B x M x K i64 -> (center B x 3 x M, de-centred neighbour coordinates B x 3 x (K*M) K-MAJOR,
    gather index B x (K*M) i32) -- what the gathering KNNModule layer needs beside the features (models/layers.py:313-350)."""
    _chk(coord, "coord", torch.float32, 3)
    _chk(knn_I, "knn_I", torch.int64, 3)
    dev = _same_device(coord, knn_I)
    B, _, M = coord.shape
    K = knn_I.shape[2]
    if coord.shape[1] != 3 or knn_I.shape[0] != B or knn_I.shape[1] != M:
        raise SonetHipError("knn_prepare: coord must be B x 3 x M and knn_I B x M x K")
    center = torch.empty((B, 3, M), dtype=torch.float32, device=dev)
    dec = torch.empty((B, 3, K * M), dtype=torch.float32, device=dev)
    gidx = torch.empty((B, K * M), dtype=torch.int32, device=dev)
    with _lib.on_device(dev), _timed("knn_prepare"):
        check(_lib.load().sonet_knn_prepare_f32(ptr(coord), ptr(knn_I), B, M, K, int(bool(center_avg)), ptr(center), ptr(dec), ptr(gidx),
                                                stream_ptr()), "sonet_knn_prepare_f32")
    return center, dec, gidx



def planes_max(x, K):
    """x B x C x (K*M) with k-major columns -> B x C x M, max over the K planes (values only, NaN-propagating)."""
    if x.dtype == torch.bfloat16:
        x = x.float()
    _chk(x, "x", torch.float32, 3)
    B, C, L = x.shape
    if K <= 0 or L % K:
        raise SonetHipError("planes_max: L=%d is not %d planes" % (L, K))
    dev = _same_device(x)
    out = torch.empty((B, C, L // K), dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    with _lib.on_device(dev), _timed("planes_max"):
        check(_lib.load().sonet_planes_max_f32(ptr(x), ptr(out), B * C, K, L // K, stream_ptr()), "sonet_planes_max_f32")
    return out


def bn_running_update_(running_mean, running_var, mean, var, momentum, unbias):
    """running = running*(1-momentum) + momentum*stat in place (variance entering as var*unbias): F.batch_norm's update."""
    for t, n in ((running_mean, "running_mean"), (running_var, "running_var"), (mean, "mean"), (var, "var")):
        _chk(t, n, torch.float32, 1)
    dev = _same_device(running_mean, running_var, mean, var)
    with _lib.on_device(dev):
        check(_lib.load().sonet_bn_running_update_f32(ptr(running_mean), ptr(running_var), ptr(mean), ptr(var), float(momentum), float(unbias),
                                                      running_mean.numel(), stream_ptr()), "sonet_bn_running_update_f32")
    # written through raw pointers: move the version counters as an in-place aten op would (caches keyed on versions must notice)
    torch.autograd.graph.increment_version(running_mean)
    torch.autograd.graph.increment_version(running_var)
