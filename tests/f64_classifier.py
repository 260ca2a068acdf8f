"""tests/f64_classifier.py -- TEST INFRASTRUCTURE: one classifier training step of the reference in float64, in plain torch autograd,
with the three arg-max pools either FREE (computed here, as the reference computes them) or FORCED (positions given).

Follows, line by line, what ``models/classifier.py:78-99`` runs with dropout off:
  Encoder.forward           models/networks.py:111-199   (query_topk util/som.py:237-269 with sorted slots, grouping :128-172,
                                                          PointResNet models/layers.py:419-432, index_max + gather :180-185,
                                                          KNNModule models/layers.py:313-367, PointNet :384-387, global max :197)
  Classifier.forward        models/networks.py:218-227   (MyLinear models/layers.py:156-166)
  nn.CrossEntropyLoss       models/classifier.py:47,93
BatchNorm in training mode (batch statistics, biased variance: F.batch_norm(training=True), models/layers.py:60-70).

Why it exists: end-to-end gradients pass three arg-max pools; ANY change of rounding flips a few of the
10^5 winners and one flip moves a first-PointNet weight gradient by ~3e-3 rel-rms, so a free-routing comparison can only be statistical.
With the ROUTING of the implementation under test forced on this float64 run, routing flips cannot explain a difference any more and
every gradient must agree to <= 1e-4 rel-rms.  The restatement itself is pinned to the reference: in free mode it reproduces the
``grad64/`` gradients and the ``route64/`` positions the unmodified reference produced in float64 (oracle/make_golden.py
golden_train_step), see tests/test_oracle_golden.py::test_f64_restatement_reproduces_the_reference_float64_run.
"""
import torch
import torch.nn.functional as F


def _bn_train(y, sd, prefix):
    return F.batch_norm(y, None, None, sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"], True, 0.1, 1e-5)


_TAKEN = {}          # the ReLU patterns the last free run took: {layer prefix: bool tensor}


def _act(y, relu, mask, key=None):
    """ReLU, free (``mask`` None) or FORCED: a given 0 / 1 pattern instead of the sign test -- the other discrete decision of the step."""
    if not relu:
        return y
    if mask is None:
        if key is not None:
            _TAKEN[key] = (y > 0).detach()
        return F.relu(y)
    return y * mask.reshape(y.shape).to(y.dtype)


def _conv(x, sd, prefix, bn, relu, masks=None):
    w = sd[prefix + ".conv.weight"]
    y = (F.conv2d if x.dim() == 4 else F.conv1d)(x, w if w.dim() == x.dim() else w.reshape(w.shape[0], w.shape[1], *([1] * (x.dim() - 2))),
                                                    sd[prefix + ".conv.bias"])
    if bn:
        y = _bn_train(y, sd, prefix)
    return _act(y, relu, None if masks is None else masks[prefix], prefix)


def _linear(x, sd, prefix, bn, relu, masks=None):
    y = F.linear(x, sd[prefix + ".linear.weight"], sd[prefix + ".linear.bias"])
    if bn:
        y = _bn_train(y, sd, prefix)
    return _act(y, relu, None if masks is None else masks["cls." + prefix], "cls." + prefix)


def leaf_params(sd, device, dtype=torch.float64):
    """state_dict (reference key names) -> {key: float64 leaf requiring grad} for the floating-point entries."""
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            out[k] = v.detach().to(device=device, dtype=dtype).clone().requires_grad_(not k.endswith(("running_mean", "running_var")))
    return out


def som_stage(pc, sn, node, k):
    """util/som.py:237-269 (slots in ascending distance order: the reference under ref_harness.sorted_topk()) + models/networks.py:128-172,
    in the dtype of the inputs.  -> min_idx B x kN (k-major), row_max B x M, som_node B x 3 x M, x_augmented B x 6 x kN."""
    B, _, N = pc.shape
    M = node.shape[2]
    diff = pc.unsqueeze(3) - node.unsqueeze(2)                                   # B x 3 x N x M
    diff_norm = (diff ** 2).sum(dim=1)
    _, idx = torch.topk(diff_norm, k=k, dim=2, largest=False, sorted=True)       # B x N x k
    min_idx = torch.cat([idx[..., i] for i in range(k)], dim=1)                  # B x kN
    mask = F.one_hot(min_idx, M)                                                 # B x kN x M
    row_max = mask.max(dim=1)[0]
    row_sum = mask.sum(dim=1)
    x_stack = torch.cat([pc] * k, dim=2)
    sn_stack = torch.cat([sn] * k, dim=2)
    fm = mask.to(pc.dtype).unsqueeze(1)                                          # B x 1 x kN x M
    som_node = (x_stack.unsqueeze(3) * fm).sum(dim=2) / (row_sum.unsqueeze(1).float() + 1e-5)   # (.float(): networks.py:142, also in a float64 run)
    centers = (fm * som_node.unsqueeze(2)).sum(dim=3)
    x_aug = torch.cat((x_stack - centers, sn_stack), dim=1)
    return min_idx, row_max, som_node, x_aug


def index_max_positions(first, min_idx, M, zero_pos=None):
    """models/index_max_ext/index_max.cpp:97-109 on float32 values (the extension reads float32 only: oracle/make_golden.py casts at the
    shim): running max starts at -1000 / position 0, strict '>', ascending n.  first B x C x L, min_idx B x L -> B x C x M (int64).
    ``zero_pos`` (B,): where the reference's "position 0" -- a bin nothing beat -- sits in THIS column order (node-sorted columns: the
    sorted position of original column 0); default 0."""
    v = first.detach().float()
    B, C, L = v.shape
    out = torch.zeros(B, C, M, dtype=torch.int64, device=v.device)
    ar = torch.arange(L, device=v.device)
    z = torch.zeros(B, 1, dtype=torch.int64, device=v.device) if zero_pos is None else zero_pos.long().view(B, 1)
    for m in range(M):
        sel = (min_idx == m)                                                     # B x L
        vm = torch.where(sel.unsqueeze(1), v, torch.full_like(v, -float("inf")))
        best = vm.max(dim=2, keepdim=True)[0]                                    # B x C x 1
        first_at = torch.where(vm == best, ar.view(1, 1, L), torch.full((1, 1, 1), L, device=v.device)).min(dim=2)[0]
        ok = (best.squeeze(2) > -1000.0) & (first_at < L)
        out[:, :, m] = torch.where(ok, first_at, z.expand(B, C))
    return out


def train_step(enc, cls, label, node_knn_I, som_k=9, pc=None, sn=None, node=None, k=3, stage=None, route=None, masks=None):
    """One forward + backward.  ``enc`` / ``cls``: ``leaf_params`` dictionaries.
    Inputs either (pc, sn, node) -- the SOM stage runs here, in their dtype -- or ``stage`` = dict(x_aug B x 6 x kN, min_idx B x kN,
    row_max B x M, som_node B x 3 x M[, pos0 B]) taken from the implementation under test (any column order: a point-wise network and
    BatchNorm's batch sums do not care; pos0 = where original column 0 sits in that order, the reference's gather position of an empty
    node, models/networks.py:185).  ``route`` = None (free) or dict(pool1 B x 384 x M positions ALREADY multiplied by row_max, pool2 B x 512 x M,
    pool3 B x F).  ``masks`` = None (ReLU as usual) or {layer prefix ("first_pointnet.layers.0" .. "cls.fc2"): 0 / 1 tensor with the
    element count of that layer's output, in the column order of the stage}: the ReLU pattern of the run under test, forced.
    -> dict(loss, feature, score, grads {key: tensor}, route {pool1, pool2, pool3}, masks {layer prefix: the ReLU pattern a free run took})."""
    dt = next(iter(enc.values())).dtype
    _TAKEN.clear()
    if stage is None:
        min_idx, row_max, som_node, x_aug = som_stage(pc.to(dt), sn.to(dt), node.to(dt), k)
    else:
        min_idx, row_max = stage["min_idx"].long(), stage["row_max"].long()
        som_node, x_aug = stage["som_node"].to(dt), stage["x_aug"].to(dt)
    x_aug, som_node = x_aug.detach(), som_node.detach()
    M = som_node.shape[2]
    p = "first_pointnet.layers."
    l0 = _conv(x_aug, enc, p + "0", True, True, masks)
    t = _conv(_conv(l0, enc, p + "1", True, True, masks), enc, p + "2", True, True, masks)
    first = _conv(torch.cat((l0, t), dim=1), enc, p + "3", False, False)          # B x 384 x kN
    if route is None:
        zero_pos = None if stage is None or stage.get("pos0") is None else stage["pos0"]
        pool1 = index_max_positions(first, min_idx, M, zero_pos)
        zp = torch.zeros(pool1.shape[0], 1, 1, dtype=torch.int64, device=pool1.device) if zero_pos is None else zero_pos.long().view(-1, 1, 1)
        pool1 = torch.where(row_max.unsqueeze(1) > 0, pool1, zp.expand_as(pool1))           # networks.py:185: index * mask_row_max
    else:
        pool1 = route["pool1"].long()
    masked_max = first.gather(2, pool1)                                           # networks.py:185
    knn_I = node_knn_I[:, :, :som_k].long()
    B = som_node.shape[0]

    def knn_gather(x):                                                            # operations.py:19-54
        C = x.shape[1]
        return x.gather(2, knn_I.reshape(B, 1, -1).expand(B, C, M * som_k)).reshape(B, C, M, som_k)
    nb = knn_gather(som_node)
    center = nb.mean(dim=3, keepdim=True)
    h = torch.cat(((nb - center).detach(), knn_gather(masked_max)), dim=1)
    h = _conv(_conv(h, enc, "knnlayer.layers.0", True, True, masks), enc, "knnlayer.layers.1", True, True, masks)   # B x 512 x M x K'
    pool2 = h.max(dim=3)[1] if route is None else route["pool2"].long()
    knn_feature = h.gather(3, pool2.unsqueeze(3)).squeeze(3)
    f = torch.cat((center.squeeze(3).detach(), knn_feature), dim=1)
    final = _conv(_conv(f, enc, "final_pointnet.layers.0", True, True, masks), enc, "final_pointnet.layers.1", False, False)
    pool3 = final.max(dim=2)[1] if route is None else route["pool3"].long()
    feature = final.gather(2, pool3.unsqueeze(2)).squeeze(2)
    s = _linear(_linear(feature, cls, "fc1", True, True, masks), cls, "fc2", True, True, masks)
    score = _linear(s, cls, "fc3", False, False)
    loss = F.cross_entropy(score, label.long())
    leaves = {k_: v for k_, v in list(enc.items()) + [("cls." + k_, v) for k_, v in cls.items()] if v.requires_grad}
    gr = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    return dict(loss=loss.detach(), feature=feature.detach(), score=score.detach(), grads={k_: g for k_, g in zip(leaves, gr) if g is not None},
                route=dict(pool1=pool1, pool2=pool2, pool3=pool3), masks=dict(_TAKEN))
