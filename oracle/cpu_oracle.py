"""oracle/cpu_oracle.py -- TEST INFRASTRUCTURE: python face of the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (so-net_amd/) never does.  It offers

* thin numpy wrappers over oracle/sonet_oracle.c (the plain-C restatement, built by build_ref.py),
* ``ref_index_max`` -- the reference's own compiled ``index_max_forward_cpu`` (oracle/_ref) when
  that binary exists,
* ``encoder_forward`` / ``classifier_forward`` -- a restatement of the reference forward
  (models/networks.py:111-199, :218-227) that chains the C functions with the very aten CPU calls
  the reference itself makes for its float layers (F.conv1d / F.batch_norm / F.linear,
  models/layers.py:68,286,157).  aten is the reference's un-pinned third-party dependency
  ("PyTorch 0.4 or higher", README.md:30); the same torch 2.10 CPU build is on the GPU box.

Weights are passed as a plain ``state_dict`` with the reference's key names.
"""
import ctypes
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f = ctypes.POINTER(ctypes.c_float)
_i32 = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "_build", "libsonet_oracle.so")
        if not os.path.exists(path):
            from . import build_ref
            build_ref.build_oracle()
        _LIB = ctypes.CDLL(path)
    return _LIB


def ref_module():
    """The reference's own index_max extension (CPU entry points), or None if not built."""
    global _REF
    if _REF is None:
        path = os.path.join(HERE, "_ref", "index_max.so")
        if not os.path.exists(path):
            return None
        import torch  # noqa: F401  (the extension links libtorch)
        spec = importlib.util.spec_from_file_location("index_max", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _REF = mod
    return _REF


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a, t):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------ C restatement wrappers
def index_max(data, index, K):
    """index_max.cpp:73-112.  data BxCxN f32, index BxN i32 -> BxCxK i32."""
    data = _c(data, np.float32)
    index = _c(index, np.int32)
    B, C, N = data.shape
    out = np.empty((B, C, K), np.int32)
    lib().oracle_index_max_f32(_p(data, _f), _p(index, _i32), _p(out, _i32), B, C, N, K)
    return out


def ref_index_max(data, index, K, threads=0):
    """The reference's compiled forward_cpu / forward_multi_thread_cpu (oracle/_ref)."""
    import torch
    m = ref_module()
    if m is None:
        raise RuntimeError("oracle/_ref/index_max.so is not built")
    d = torch.from_numpy(_c(data, np.float32))
    i = torch.from_numpy(_c(index, np.int32))
    if threads and threads > 1:
        return m.forward_multi_thread_cpu(d, i, int(K), int(threads)).numpy()
    return m.forward_cpu(d, i, int(K)).numpy()


def som_query_topk(x, node, k):
    """util/som.py:237-269 -> (min_idx BxkN i64 k-major canonical order, count BxM i32, row_max BxM i32)."""
    x = _c(x, np.float32)
    node = _c(node, np.float32)
    B, _, N = x.shape
    M = node.shape[2]
    min_idx = np.empty((B, k * N), np.int64)
    count = np.empty((B, M), np.int32)
    row_max = np.empty((B, M), np.int32)
    lib().oracle_som_query_topk_f32(_p(x, _f), _p(node, _f), B, N, M, k,
                                    _p(min_idx, _i64), _p(count, _i32), _p(row_max, _i32))
    return min_idx, count, row_max


def mask_from_min_idx(min_idx, M):
    """one-hot BxkNxM i32 (util/som.py:254-265)."""
    return (min_idx[..., None] == np.arange(M, dtype=np.int64)).astype(np.int32)


def som_group(x, min_idx, M, k):
    """models/networks.py:128-171 -> (som_node Bx3xM, centers Bx3xkN, x_decentered Bx3xkN)."""
    x = _c(x, np.float32)
    min_idx = _c(min_idx, np.int64)
    B, _, N = x.shape
    som_node = np.empty((B, 3, M), np.float32)
    centers = np.empty((B, 3, k * N), np.float32)
    xd = np.empty((B, 3, k * N), np.float32)
    lib().oracle_som_group_f32(_p(x, _f), _p(min_idx, _i64), B, N, M, k,
                               _p(som_node, _f), _p(centers, _f), _p(xd, _f))
    return som_node, centers, xd


def knn_gather(x, I):
    """models/operations.py:38-54."""
    x = _c(x, np.float32)
    I = _c(I, np.int64)
    B, C, M = x.shape
    K = I.shape[2]
    out = np.empty((B, C, M, K), np.float32)
    lib().oracle_knn_gather_f32(_p(x, _f), _p(I, _i64), _p(out, _f), B, C, M, K)
    return out


def pointwise_layer(x, W, bias, bn=None, relu=True, eps=1e-5):
    """models/layers.py:282-296, eval-mode BN.  bn = (gamma, beta, running_mean, running_var) or None."""
    x = _c(x, np.float32)
    B, Cin, L = x.shape
    W = _c(np.asarray(W).reshape(-1, Cin), np.float32)
    Cout = W.shape[0]
    bias = _c(bias, np.float32)
    if bn is None:
        g = b_ = m = v = np.zeros(Cout, np.float32)
    else:
        g, b_, m, v = (_c(t, np.float32) for t in bn)
    y = np.empty((B, Cout, L), np.float32)
    lib().oracle_pointwise_layer_f32(_p(x, _f), _p(W, _f), _p(bias, _f), _p(g, _f), _p(b_, _f),
                                     _p(m, _f), _p(v, _f), ctypes.c_float(eps),
                                     int(bn is not None), int(bool(relu)), _p(y, _f),
                                     B, Cin, Cout, L)
    return y


def chamfer_nn(q, db):
    """models/losses.py:220-235 (faiss IndexFlatL2 1-NN, restated; parity unpinned)."""
    q = _c(q, np.float32)
    db = _c(db, np.float32)
    B, _, Nq = q.shape
    Nd = db.shape[2]
    nn = np.empty((B, Nq), np.int32)
    lib().oracle_chamfer_nn_f32(_p(q, _f), _p(db, _f), _p(nn, _i32), B, Nq, Nd)
    return nn


def chamfer_loss(predict, gt):
    """ChamferLoss.forward restated (models/losses.py:237-290) on numpy float32: exact 1-NN both ways (chamfer_nn
    above), gather, robust_norm = sqrt(sum_xyz(d^2) + 1e-8) (:17-22), means.  Returns (forward_loss, backward_loss,
    loss_array B)."""
    predict = _c(predict, np.float32)
    gt = _c(gt, np.float32)
    nn_gt = chamfer_nn(predict, gt)                    # predicted -> nearest gt point
    nn_pr = chamfer_nn(gt, predict)                    # gt -> nearest predicted point
    sel_gt = np.take_along_axis(gt, nn_gt[:, None, :].astype(np.int64).repeat(3, 1), axis=2)
    sel_pr = np.take_along_axis(predict, nn_pr[:, None, :].astype(np.int64).repeat(3, 1), axis=2)
    fwd = np.sqrt(((sel_gt - predict) ** 2).sum(1, dtype=np.float32) + np.float32(1e-8))      # B x M
    bwd = np.sqrt(((sel_pr - gt) ** 2).sum(1, dtype=np.float32) + np.float32(1e-8))           # B x N
    return float(fwd.mean(dtype=np.float64)), float(bwd.mean(dtype=np.float64)), fwd.mean(1) + bwd.mean(1)


# ------------------------------------------------------------------ forward restatement (torch CPU)
def _eq_layer(sd, prefix, x, bn, relu, eps=1e-5, conv2d=False):
    """EquivariantLayer / MyConv2d forward in eval mode (models/layers.py:282-296, :199-211)."""
    import torch.nn.functional as F
    w, b = sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"]
    y = F.conv2d(x, w, b) if conv2d else F.conv1d(x, w, b)
    if bn:
        y = F.batch_norm(y, sd[prefix + ".norm.running_mean"], sd[prefix + ".norm.running_var"],
                         sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"], False, 0.1, eps)
    if relu:
        y = F.relu(y)
    return y


def _linear(sd, prefix, x, bn, relu, eps=1e-5):
    """MyLinear forward in eval mode (models/layers.py:156-166)."""
    import torch.nn.functional as F
    y = F.linear(x, sd[prefix + ".linear.weight"], sd[prefix + ".linear.bias"])
    if bn:
        y = F.batch_norm(y, sd[prefix + ".norm.running_mean"], sd[prefix + ".norm.running_var"],
                         sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"], False, 0.1, eps)
    if relu:
        y = F.relu(y)
    return y


def encoder_forward(sd, x, sn, node, node_knn_I, k=3, som_k=9, som_k_type="avg",
                    index_max_threads=0, use_ref_index_max=True):
    """Encoder.forward, eval mode, surface_normal=True, som_k>=2 (models/networks.py:111-199).

    sd: encoder state_dict (reference key names, CPU f32 tensors).  Inputs are CPU torch tensors.
    Returns a dict of every intermediate the heads read.
    """
    import torch
    B, _, N = x.shape
    M = node.shape[2]
    min_idx, count, row_max = som_query_topk(x.numpy(), node.numpy(), k)          # networks.py:127
    som_node, centers, xd = som_group(x.numpy(), min_idx, M, k)                   # :128-171
    sn_stack = torch.cat([sn] * k, dim=2)                                          # :132-137
    x_aug = torch.cat((torch.from_numpy(xd), sn_stack), dim=1)                     # :172
    # first PointNet = PointResNet(6,[64,128,256,384])  (networks.py:82, layers.py:419-432)
    p = "first_pointnet.layers."
    l0 = _eq_layer(sd, p + "0", x_aug, True, True)
    t = _eq_layer(sd, p + "1", l0, True, True)
    t = _eq_layer(sd, p + "2", t, True, True)
    first = _eq_layer(sd, p + "3", torch.cat((l0, t), dim=1), False, False)
    # index_max (networks.py:180-184) + masked gather (:185)
    idx32 = min_idx.astype(np.int32)
    if use_ref_index_max and ref_module() is not None:
        gi = ref_index_max(first.numpy(), idx32, M, threads=index_max_threads)
    else:
        gi = index_max(first.numpy(), idx32, M)
    gather_index = torch.from_numpy(gi.astype(np.int64)) * torch.from_numpy(row_max.astype(np.int64)).unsqueeze(1)
    masked_max = first.gather(2, gather_index)
    # KNNModule (layers.py:313-367)
    node_t = torch.from_numpy(som_node)
    knn_I = node_knn_I[:, :, :som_k]
    nb = torch.from_numpy(knn_gather(som_node, knn_I.numpy()))                     # operations.py:19
    if som_k_type == "avg":
        nb_center = nb.mean(dim=3, keepdim=True)
    else:
        nb_center = node_t.unsqueeze(3)
    nb_dec = nb - nb_center
    xnb = torch.from_numpy(knn_gather(masked_max.numpy(), knn_I.numpy()))
    h = torch.cat((nb_dec, xnb), dim=1)
    h = _eq_layer(sd, "knnlayer.layers.0", h, True, True, conv2d=True)
    h = _eq_layer(sd, "knnlayer.layers.1", h, True, True, conv2d=True)
    knn_feature, _ = h.max(dim=3)
    knn_center = nb_center.squeeze(3)
    # final PointNet(515 -> 768 -> feature_num) (networks.py:94,192) + global max (:197)
    f = torch.cat((knn_center, knn_feature), dim=1)
    f = _eq_layer(sd, "final_pointnet.layers.0", f, True, True)
    final = _eq_layer(sd, "final_pointnet.layers.1", f, False, False)
    feature, _ = final.max(dim=2)
    return dict(min_idx=min_idx, count=count, row_max=row_max, som_node=som_node, centers=centers,
                x_decentered=xd, first_pn_out=first, gather_index=gi, first_pn_out_masked_max=masked_max,
                knn_center_1=knn_center, knn_feature_1=knn_feature, final_pn_out=final, feature=feature)


def classifier_forward(sd, feature):
    """Classifier.forward, eval mode (models/networks.py:218-227; dropout is identity in eval)."""
    h = _linear(sd, "fc1", feature, True, True)
    h = _linear(sd, "fc2", h, True, True)
    return _linear(sd, "fc3", h, False, False)
