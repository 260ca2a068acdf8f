"""Level-2 drop-in of the reference's models/networks.py: ``Encoder`` (the hot path) and the
``Classifier`` head used by the ModelNet40 benchmark.

``Encoder(opt)`` keeps the reference constructor, ``forward(x, sn, node, node_knn_I, is_train, epoch)
-> B x feature_num``, the post-call attributes the heads read (``mask`` [lazy], ``som_node``,
``centers``, ``x_decentered``, ``first_pn_out``, ``first_pn_out_masked_max``, ``knn_center_1``,
``knn_feature_1``, ``final_pn_out``, ``feature``, ``som_builder.node``; models/segmenter.py:90-109)
and the ``state_dict`` keys -- including the never-called ``transformer.*`` (models/networks.py:78,
147-164) -- so reference checkpoints load and the Classifier / Segmenter / Decoder heads of a
reference checkout run on top unchanged ("two drop-in levels").

What changes is the data flow of models/networks.py:111-199:

  reference                                              here
  ---------------------------------------------------   ------------------------------------------
  query_topk -> dense one-hot mask B x kN x M            som_assign kernel: int32 node ids + per-node
  masked broadcast-multiply-sum for the cluster mean      counts / f64 coordinate sums (LDS atomics)
  (two B x 3 x kN x M f32 temporaries, :141, :169)        som_group kernel: mean, de-centre, concat sn
  4 x (conv1d, batch_norm, relu) + cat                    4 fused pointmlp launches (MFMA), skip concat
                                                          read in place
  index_max ext + torch.gather                            index_max_gather kernel (arg-max + value)
  KNNModule / final PointNet on aten                      same modules on the fused kernels
"""
import math

import weakref

import torch
import torch.nn as nn

from sonet_hip import ops as _ops
from util import som
from .layers import EquivariantLayer, KNNModule, MyLinear, PointNet, PointResNet, _PlainAttrs  # noqa: F401


def _head_inference(fn):
    """Head ``forward`` wrapper: an eval-mode head fed with the tagged outputs of an inference encoder call
    (``Encoder.forward(is_train=False)``, sonet_hip.ops.mark_inference) runs under ``torch.no_grad()`` even though the
    reference's ``Model.test_model`` leaves autograd on (models/classifier.py:101-105) -- that is what selects the fused
    no-autograd kernels of the layers underneath."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        ts = [a for a in args if isinstance(a, torch.Tensor) and a.is_floating_point()]
        dev = ts[0].device if ts else torch.device("cpu")
        # (``head.inference = False`` opts a head out: an eval-mode head with trainable weights fed with frozen encoder features
        # then keeps its autograd graph, as in the reference)
        if torch.is_grad_enabled() and not self.training and getattr(self, "inference", True):
            if ts and not any(t.requires_grad for t in ts) and any(getattr(t, "_sonet_inference", False) for t in ts):
                with torch.no_grad():
                    return _ops.mark_inference(_ops.run_guarded(lambda: fn(self, *args, **kwargs), dev, True))
        # (the heads' own h3 launches -- the segmenter's point-wise layers, the decoder's 1x1 convs -- are range-guarded too)
        return _ops.run_guarded(lambda: fn(self, *args, **kwargs), dev, not (torch.is_grad_enabled() or self.training))
    return wrapper


def _bn_kwargs(opt):
    return dict(momentum=opt.bn_momentum, bn_momentum_decay_step=opt.bn_momentum_decay_step,
                bn_momentum_decay=opt.bn_momentum_decay)


class Transformer(nn.Module):
    """Rotation regressor that the reference constructs but never calls (models/networks.py:20-68,
    forward use commented out at :147-164).  Kept so that ``state_dict`` keys match; 184,449
    parameters that never receive a gradient (the data-parallel wrapper skips them)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        kw = _bn_kwargs(opt)
        self.first_pointnet = PointNet(3, (32, 64, 128), activation=opt.activation, normalization=opt.normalization, **kw)
        self.second_pointnet = PointNet(128 + 128, (256, 256), activation=opt.activation, normalization=opt.normalization, **kw)
        self.fc1 = MyLinear(256, 128, activation=opt.activation, normalization=opt.normalization, **kw)
        self.fc2 = MyLinear(128, 64, activation=opt.activation, normalization=opt.normalization, **kw)
        self.fc3 = MyLinear(64, 1, activation=None, normalization=None)
        self.dropout1 = nn.Dropout(p=opt.dropout)
        self.dropout2 = nn.Dropout(p=opt.dropout)

    def forward(self, x, sn=None, epoch=None):
        h = self.first_pointnet(x, epoch)
        g1, _ = torch.max(h, dim=2, keepdim=True)
        h = self.second_pointnet(torch.cat((h, g1.expand_as(h)), dim=1), epoch)
        g2, _ = torch.max(h, dim=2)
        h = self.fc1(g2, epoch)
        if self.opt.dropout > 0.1:
            h = self.dropout1(h)
        self.fc2_out = self.fc2(h, epoch)
        if self.opt.dropout > 0.1:
            self.fc2_out = self.dropout2(self.fc2_out)
        return torch.tanh(self.fc3(self.fc2_out, epoch))


# Live Segmenter instances of this process: their Model (models/segmenter.py:79-109) reads encoder.first_pn_out per point copy without
# telling the encoder; a training forward keeps the tensor then (see Encoder._wants_dense; ``encoder.want_first_pn_out = False`` overrides)
_DENSE_HEADS = weakref.WeakSet()


class Encoder(_PlainAttrs, nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.feature_num = opt.feature_num
        kw = _bn_kwargs(opt)
        act, norm = opt.activation, opt.normalization
        self.transformer = Transformer(opt)
        self.first_pointnet = PointResNet(6 if opt.surface_normal else 3, [64, 128, 256, 384],
                                          activation=act, normalization=norm, **kw)
        if opt.som_k >= 2:
            self.knnlayer = KNNModule(3 + 384, (512, 512), activation=act, normalization=norm, **kw)
            self.final_pointnet = PointNet(3 + 512, (768, self.feature_num), activation=act, normalization=norm, **kw)
        else:
            self.final_pointnet = PointResNet(3 + 384, (512, 512, 768, self.feature_num),
                                              activation=act, normalization=norm, **kw)
        rows = int(math.sqrt(opt.node_num))
        self.som_builder = som.BatchSOM(rows, rows, 3, opt.gpu_id, opt.batch_size)
        self.zero_pad = torch.nn.ZeroPad2d(padding=1)
        self._lazy = None
        self._first_pn_out = None

    # ---- attributes the segmenter / autoencoder read after forward; built on demand -----------------
    @property
    def mask(self):
        """B x kN x M int32 one-hot assignment (models/segmenter.py:90 takes its argmax)."""
        st = self._lazy
        if st is None:
            raise AttributeError("Encoder.mask: call forward first")
        if st.get("mask") is None:
            st["mask"] = _ops.som_mask(st["a"].min_idx_i32, st["a"].M)
        return st["mask"]

    @property
    def min_idx(self):
        """B x kN int64 node id of every point copy (what the segmenter recovers with argmax(mask))."""
        st = self._lazy
        if st.get("min_idx") is None:
            st["min_idx"] = st["a"].min_idx_i32.long()
        return st["min_idx"]

    def _per_point(self, name):
        st = self._lazy
        if st is None:
            raise AttributeError("Encoder.%s: call forward first" % name)
        if st.get(name) is None:
            g = _ops.som_group(st["x"], None, st["a"], want_centers=True, want_decentered=True)
            st["centers"], st["x_decentered"] = g["centers"], g["x_decentered"]
        return _ops.mark_inference(st[name]) if getattr(self, "_infer_tag", False) else st[name]

    @property
    def first_pn_out(self):
        """B x 384 x kN output of the first PointNet.  On the fused no-grad path it is never written during
        forward (only its per-node max is needed); it is materialised here if a caller (the segmenter) reads it."""
        if self._first_pn_out is None and self._lazy is not None and self._lazy.get("first_p16") is not None:
            self._first_pn_out = self._lazy["first_p16"].float()
        if self._first_pn_out is None and self._lazy is not None and self._lazy.get("not_materialised"):
            # (RuntimeError, not AttributeError: nn.Module.__getattr__ is the fall-back of a property that raises AttributeError and would
            #  replace this message by "'Encoder' object has no attribute 'first_pn_out'" -- and make hasattr() / getattr(.., None) lie)
            raise RuntimeError("Encoder.first_pn_out was not materialised by this training forward (only its per-node maximum is consumed: the "
                               "pool ran in the last layer's epilogue); set encoder.want_first_pn_out = True before the forward to keep it")
        if self._first_pn_out is None and self._lazy is not None:
            st = self._lazy
            g = _ops.som_group(st["x"], st["sn"], st["a"], want_augmented=st["sn"] is not None, want_decentered=st["sn"] is None)
            with torch.no_grad():
                self._first_pn_out = self.first_pointnet(g["x_augmented"] if st["sn"] is not None else g["x_decentered"], None)
        if getattr(self, "_infer_tag", False):
            _ops.mark_inference(self._first_pn_out)
        return self._first_pn_out

    @first_pn_out.setter
    def first_pn_out(self, v):
        self._first_pn_out = v

    # (the flat node-level stage keeps its intermediate maps pre-split; the f32 tensors the reference leaves on the module are built
    #  when somebody reads them)
    def _stage_map(self, name):
        v = self.__dict__.get("_" + name)
        st = self.__dict__.get("_stage")
        if v is None and st is not None:
            with torch.no_grad():
                if name == "knn_feature_1":
                    v = _ops.p16_flat_to_bcm(st["knn_p16"], st["B"], st["M"])
                else:                                   # final_pn_out: the last layer once more, this time with its columns written
                    lyr = self.final_pointnet.layers[1]
                    sc, sh = lyr._eval_affine()
                    y = _ops.pointmlp_h3p(st["h3_p16"], lyr._packed_p16(), sc, sh, lyr.activation == 'relu', lyr.conv.out_channels, out="f32")
                    v = y[0, :, :st["B"] * st["M"]].reshape(-1, st["B"], st["M"]).permute(1, 0, 2).contiguous()
            if getattr(self, "_infer_tag", False):
                _ops.mark_inference(v)
            self.__dict__["_" + name] = v
        return v

    @property
    def knn_feature_1(self):
        return self._stage_map("knn_feature_1")

    @knn_feature_1.setter
    def knn_feature_1(self, v):
        self.__dict__["_knn_feature_1"] = v

    @property
    def final_pn_out(self):
        return self._stage_map("final_pn_out")

    @final_pn_out.setter
    def final_pn_out(self, v):
        self.__dict__["_final_pn_out"] = v

    @property
    def centers(self):
        return self._per_point("centers")

    @property
    def x_decentered(self):
        return self._per_point("x_decentered")

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, sn, node, node_knn_I, is_train=False, epoch=None):
        """x, sn: B x 3 x N; node: B x 3 x M; node_knn_I: B x M x K' int64 -> B x feature_num.

        ``is_train`` is the reference's own signal (models/networks.py:111; ``Model.test_model`` passes False,
        ``Model.optimize`` True -- models/classifier.py:90,104): ``is_train=False`` on an ``eval()`` encoder is an inference
        call, and runs under ``torch.no_grad()`` even when the caller did not say so (the reference's ``test_model`` does
        not), which is what selects the fused no-autograd kernels.  The outputs are tagged so that the heads that consume
        them (Classifier / Segmenter / Decoder layers in eval mode) do the same.  ``encoder.inference = False`` restores
        autograd for eval-mode forwards (e.g. fine-tuning with frozen BatchNorm statistics through ``is_train=False``)."""
        infer = (not is_train) and (not self.training) and getattr(self, "inference", True) and not x.requires_grad
        if infer and torch.is_grad_enabled():
            with torch.no_grad():
                out = self._forward_guarded(x, sn, node, node_knn_I, is_train, epoch)
        else:
            out = self._forward_guarded(x, sn, node, node_knn_I, is_train, epoch)
        self._infer_tag = (not torch.is_grad_enabled()) or infer
        if self._infer_tag:
            # (the raw slots, not the properties: a map the flat node-level stage kept pre-split is decoded -- and tagged -- when it is read)
            for t in (out, self.first_pn_out_masked_max, self.__dict__.get("_final_pn_out"), self.som_node, self.__dict__.get("_knn_feature_1")):
                _ops.mark_inference(t)
        return out

    def _wants_dense(self):
        """Does a head read first_pn_out per point copy?  ``want_first_pn_out`` when somebody set it, else: does a Segmenter exist."""
        v = self.__dict__.get('want_first_pn_out')
        return bool(v) if v is not None else len(_DENSE_HEADS) > 0

    # ---- no-grad node-level stage on the flat column axis (csrc/node_stage.hip) ---------------------------------------------------
    def _node_stage_ok(self, B, M, node_knn_I):
        """KNNModule + final PointNet + global max as five launches on pre-split activations: eval, no autograd, h3 arithmetic, the
        standard two-layer modules, 64 or 128 nodes (a cloud's nodes = one or half a 128-column block)."""
        opt = self.opt
        if not (_ops.NODE_STAGE_P16 and _ops.GATHER_NODE_STAGE and _ops.P16_CHAINS and _ops.POINTMLP_PRECISION == "h3" and opt.som_k >= 2) \
                or torch.is_grad_enabled():
            return False
        if getattr(self, 'want_first_pn_out', False):           # (a head that reads the intermediate maps densely: the segmenter)
            return False
        if M not in (64, 128) or node_knn_I is None or node_knn_I.dim() != 3 or node_knn_I.dtype != torch.int64 \
                or node_knn_I.shape[2] < opt.som_k or opt.som_k > 128 or opt.som_k_type not in ('avg', 'center'):
            return False
        if not isinstance(self.final_pointnet, PointNet) or len(self.knnlayer.layers) != 2 or len(self.final_pointnet.layers) != 2:
            return False
        k1, k2 = self.knnlayer.layers
        f1, f2 = self.final_pointnet.layers
        if k1.conv.in_channels != 3 + 384 or k2.conv.in_channels != k1.conv.out_channels or k1.conv.out_channels % 16 != 0 \
                or f1.conv.in_channels != 3 + k2.conv.out_channels or k2.conv.out_channels % 16 != 0 or f2.conv.in_channels != f1.conv.out_channels:
            return False
        return all(l._p16_ok() for l in (k1, k2, f1, f2))

    def _node_stage(self, xp, prep, B, M):
        """xp: ``P16`` 1 x 384 x Lm (the pooled map from the fused first PointNet's decode pass), prep: ``knn_stage_prepare`` of the
        nodes (launched before the first PointNet: it needs the coordinates only) -> feature B x feature_num."""
        K, G = prep["K"], prep["G"]
        k1, k2 = self.knnlayer.layers
        f1, f2 = self.final_pointnet.layers
        dev = xp.device
        C1 = k1.conv.out_channels
        # KNNModule layer 1 = (its 384-channel block once per node) + (gather + the 3 coordinate channels per neighbour copy)
        z = _ops.pointmlp_h3p(xp, k1._packed_p16(lambda: k1._weight2d().detach()[:, 3:], "feat"), _ops.const_vec(C1, 1.0, dev),
                              _ops.const_vec(C1, 0.0, dev), False, C1, out="p16", tag="flat")
        s1, t1 = k1._eval_affine()
        h1 = _ops.knn_stage_input(prep, z, k1._lead_cols(3), s1, t1, k1.activation == 'relu')
        center, cp = prep["center"], prep["center_p16"]
        # layer 2 + max over the K neighbours (models/layers.py:352-365)
        s2, t2 = k2._eval_affine()
        knn = _ops.pointmlp_h3p_gmax(h1, k2._packed_p16(), s2, t2, k2.activation == 'relu', k2.conv.out_channels, K, G, B * M, out="p16", Lout=xp.L)
        # final PointNet on cat(center, knn feature) (models/networks.py:191-196): the 3 centre channels as the second panel
        s3, t3 = f1._eval_affine()
        h3 = _ops.pointmlp_h3p(knn, f1._packed_p16(lambda: torch.cat((f1._weight2d().detach()[:, 3:], f1._weight2d().detach()[:, :3]), dim=1), "rot3"),
                               s3, t3, f1.activation == 'relu', f1.conv.out_channels, x2=cp, out="p16", tag="flat")
        # last layer + max over the cloud's M nodes (models/networks.py:197)
        s4, t4 = f2._eval_affine()
        feature = _ops.pointmlp_h3p_gmax(h3, f2._packed_p16(), s4, t4, f2.activation == 'relu', f2.conv.out_channels, M, 128 // M, B, out="f32")
        self.knn_center_1 = center
        self.__dict__["_stage"] = dict(knn_p16=knn, h3_p16=h3, B=B, M=M)
        self.__dict__["_knn_feature_1"] = None
        self.__dict__["_final_pn_out"] = None
        return feature

    def _forward_guarded(self, x, sn, node, node_knn_I, is_train, epoch):
        """The forward inside an operand-range scope of the fp16-split arithmetic (sonet_hip/ops.py ``run_guarded``):
        no-grad calls are recomputed in the range-safe x3 arithmetic when a launch left the fp16 range; training calls
        (BatchNorm statistics: must not run twice) are checked one step late and switch the process to x3."""
        can_rerun = not (torch.is_grad_enabled() or self.training)
        return _ops.run_guarded(lambda: self._forward(x, sn, node, node_knn_I, is_train, epoch), x.device, can_rerun)

    def _forward(self, x, sn, node, node_knn_I, is_train=False, epoch=None):
        opt = self.opt
        M = node.size()[2]
        xd = x.detach().float().contiguous()
        sb = self.som_builder
        sb.node = node.detach().float().contiguous()                     # networks.py:124
        use_sn = bool(opt.surface_normal)
        snd = sn.detach().float().contiguous() if use_sn else None
        # (a head that reads first_pn_out densely -- the segmenter -- sets want_first_pn_out: one store-variant pass then)
        fused_pool = (use_sn and _ops.FUSE_POOL and not getattr(self, 'want_first_pn_out', False)
                      and not torch.is_grad_enabled() and self.first_pointnet._fusable_eval(xd)
                      and int(opt.k) * xd.shape[2] * 384 * 4 < 4e9)
        # the flat node-level stage (KNNModule + final PointNet + global max on pre-split activations): its index / coordinate side needs
        # the cluster means only and rides on the SOM stage's second launch
        stage = fused_pool and _ops.POINTMLP_PRECISION == "h3" and self._node_stage_ok(xd.shape[0], M, node_knn_I)
        knn = (node_knn_I.contiguous(), int(opt.som_k), opt.som_k_type == 'avg') if stage else None
        fast = sb.assign_sort(xd, snd, opt.k, knn=knn) if fused_pool else None   # no-grad fast path: assignment + node-sorted grouping in two launches (:127-172) ...
        if fast is not None:
            a, g = fast
        elif fused_pool:
            a = sb.assign(xd, opt.k)                                     # :127-128 (ids, counts, sums)
            g = _ops.som_sort_group(xd, snd, a)
        else:
            a, g = None, None                                            # (the branch below decides which grouping it wants first)
        if fused_pool:
            # ... -> first PointNet + per-node max-pool in ONE kernel (:175-185)
            sb.node = g["som_node"]
            self.som_node = sb.node
            self._lazy = dict(a=a, x=xd, sn=snd, mask=None, min_idx=None, centers=None, x_decentered=None)
            self._first_pn_out = None                                    # lazy (property)
            wstream, affine = self.first_pointnet._fused_state()
            self.__dict__["_stage"] = None
            if stage:
                # ... -> KNNModule + final PointNet + global max on the flat column axis, pre-split activations end to end (:187-197)
                prep = g.get("knn_prep")
                if prep is None:                                # (the two-launch SOM stage did not take the batch: a launch of its own)
                    prep = _ops.knn_stage_prepare(self.som_node, knn[0], knn[1], knn[2])
                self.first_pn_out_masked_max, xp = _ops.pointresnet_fused_pool(g, wstream, affine, M, want_p16=True)
                self.feature = self._node_stage(xp, prep, xd.shape[0], M)
                return self.feature
            pool = _ops.pointresnet_bf16_pool if _ops.POINTMLP_PRECISION == "bf16" else _ops.pointresnet_fused_pool
            self.first_pn_out_masked_max = pool(g, wstream, affine, M)
        else:
            self.__dict__["_stage"] = None
            # (a head that reads the per-point attributes afterwards -- the segmenter -- gets them from this launch: a second som_group
            #  launch for x_decentered / centers was 1 % of the segmenter's step)
            train_pooled = torch.is_grad_enabled() and isinstance(self.first_pointnet, PointResNet) and getattr(self, "pooled_backward", True)
            # who reads first_pn_out densely?  A head that says so (``want_first_pn_out``: segmentation_forward does) -- or, when nobody said
            # anything and autograd is on, any Segmenter that exists in this process (the reference's own models/segmenter.py Model reads
            # encoder.first_pn_out after the forward without announcing it: keep its training path whole)
            per_point = self._wants_dense() if torch.is_grad_enabled() else bool(getattr(self, 'want_first_pn_out', False))
            # f32-class training, nobody reads first_pn_out: the first PointNet runs on the NODE-SORTED copy of the points (a point-wise
            # network does not care about the column order; BatchNorm sums the same values) and its last layer pools its own output
            sorted_pool = (train_pooled and use_sn and not per_point and a is None and self.first_pointnet.pooled_sorted_ok(xd, M)
                           and int(opt.k) * xd.shape[2] * 384 * 4 < 4e9)
            if sorted_pool:
                # (two launches; the sorted position of a copy = node offset + run start + rank, the rank in a fixed order: BatchNorm's batch
                #  sums over the sorted columns are the same in every run)
                fast = sb.assign_sort(xd, snd, opt.k, deterministic=True) if _ops.TRAIN_ASSIGN_SORT else None
                if fast is not None:
                    a, g = fast
                else:
                    a = sb.assign(xd, opt.k)
                    g = _ops.som_sort_group(xd, snd, a)
                self._lazy = dict(a=a, x=xd, sn=snd, mask=None, min_idx=None, centers=None, x_decentered=None)
                pn_in = g["x_aug_sorted"]
            else:
                if a is None:
                    a = sb.assign(xd, opt.k)                                 # :127-128 (ids, counts, sums)
                g = _ops.som_group(xd, snd, a, want_centers=per_point, want_decentered=(not use_sn) or per_point, want_augmented=use_sn)   # :140-172
                self._lazy = dict(a=a, x=xd, sn=snd, mask=None, min_idx=None,
                                  centers=g["centers"], x_decentered=g["x_decentered"])
                pn_in = g["x_augmented"] if use_sn else g["x_decentered"]
            sb.node = g["som_node"]                                          # :143 cluster mean replaces the nodes
            self.som_node = sb.node
            row_max = g["row_max"]

            pooled = None
            if sorted_pool:
                pooled = self.first_pointnet.forward_pooled(pn_in, g["ids_sorted"], row_max, M, epoch, need_dense=False, pos0=g["pos0"])
            elif train_pooled:
                # training: last layer + arg-max pool as one autograd node (sparse dgrad when only the pooled output is consumed)
                # (a head that reads first_pn_out densely -- the segmenter -- sets want_first_pn_out; otherwise the tensor is not needed and,
                #  where the arithmetic has the kernel for it, never written: the pool is the last layer's epilogue)
                pooled = self.first_pointnet.forward_pooled(pn_in, a.min_idx_i32, row_max, M, epoch,
                                                            need_dense=per_point or not _ops.POOLED_TRAIN_EPILOGUE)
            if pooled is not None:
                self.first_pn_out, self.first_pn_out_masked_max, _ = pooled
                if self._first_pn_out is None:
                    self._lazy["not_materialised"] = True         # (training: recomputing it would run the BatchNorm updates twice)
            else:
                self.first_pn_out = self.first_pointnet(pn_in, epoch)        # :175-178  B x 384 x kN
                if self._first_pn_out is None:
                    # the fused first PointNet wrote only the P16 planes (emit_p16 = "only"): the pool runs on them, first_pn_out is
                    # decoded on demand (property)
                    self._lazy["first_p16"] = self.first_pointnet.last_p16
                    _, self.first_pn_out_masked_max = _ops.index_max_gather_p16(self.first_pointnet.last_p16, a.min_idx_i32, M, row_max)
                elif torch.is_grad_enabled() and self.first_pn_out.requires_grad:
                    gather_index = _ops.index_max(self.first_pn_out.detach(), a.min_idx_i32, M).long()   # :180-184
                    self.first_pn_out_masked_max = self.first_pn_out.gather(
                        dim=2, index=gather_index * row_max.unsqueeze(1).long())                         # :185
                else:
                    _, self.first_pn_out_masked_max = _ops.index_max_gather(self.first_pn_out, a.min_idx_i32, M, row_max)

        if opt.som_k >= 2:
            self.knn_center_1, self.knn_feature_1 = self.knnlayer(self.som_node, self.first_pn_out_masked_max,
                                                                  node_knn_I, opt.som_k, opt.som_k_type, epoch)
            lead, feat = self.knn_center_1, self.knn_feature_1
        else:
            lead, feat = self.som_node, self.first_pn_out_masked_max
        forward_cat = getattr(self.final_pointnet, 'forward_cat', None)   # PointNet: concat-free (rotated weight pack)
        self.final_pn_out = (forward_cat(lead, feat, epoch) if forward_cat is not None
                             else self.final_pointnet(torch.cat((lead, feat), dim=1), epoch))
        if torch.is_grad_enabled() and self.final_pn_out.requires_grad:
            self.feature = _ops.lastdim_max_autograd(self.final_pn_out)              # :197 torch.max: the gradient goes to ONE arg-max (amax would split it over ties)
        else:
            self.feature = _ops.lastdim_max(self.final_pn_out.contiguous())
        if self.feature.dtype != torch.float32:
            self.feature = self.feature.float()                                       # bf16 storage ends here: the heads' FC layers are f32
        return self.feature


class Classifier(_PlainAttrs, nn.Module):
    """feature_num -> 512 -> 256 -> classes (models/networks.py:202-227).  Three B x C FC layers:
    not part of the hot path, plain PyTorch-ROCm."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.feature_num = opt.feature_num
        kw = _bn_kwargs(opt)
        self.fc1 = MyLinear(self.feature_num, 512, activation=opt.activation, normalization=opt.normalization, **kw)
        self.fc2 = MyLinear(512, 256, activation=opt.activation, normalization=opt.normalization, **kw)
        self.fc3 = MyLinear(256, opt.classes, activation=None, normalization=None)
        self.dropout1 = nn.Dropout(p=opt.dropout)
        self.dropout2 = nn.Dropout(p=opt.dropout)

    @_head_inference
    def forward(self, feature, epoch=None):
        h = self.fc1(feature, epoch)
        if self.opt.dropout > 0.1:
            h = self.dropout1(h)
        self.fc2_out = self.fc2(h, epoch)
        if self.opt.dropout > 0.1:
            self.fc2_out = self.dropout2(self.fc2_out)
        return self.fc3(self.fc2_out, epoch)


class Segmenter(_PlainAttrs, nn.Module):
    """Per-point part-segmentation head (models/networks.py:230-344): five EquivariantLayers on the
    3356-channel concat of per-point, per-node (broadcast back) and global features; the k copies of a
    point are averaged after layer 3.  All layers run on the fused point-wise kernels."""

    def __init__(self, opt):
        super().__init__()
        _DENSE_HEADS.add(self)                            # (Encoder._wants_dense)
        self.opt = opt
        self.feature_num = opt.feature_num
        c = 3 + 3 + 3 + 16 + 384 + 384 + self.feature_num * 2
        if opt.surface_normal:
            c += 3
        if opt.som_k >= 2:
            c += 512
        act, norm = opt.activation, opt.normalization
        self.layer1 = EquivariantLayer(c, 1024, activation=act, normalization=norm)
        self.layer2 = EquivariantLayer(1024, 512, activation=act, normalization=norm)
        self.layer3 = EquivariantLayer(512, 256, activation=act, normalization=norm)
        self.drop3 = nn.Dropout(p=opt.dropout)
        self.layer4 = EquivariantLayer(256, 128, activation=act, normalization=norm)
        self.drop4 = nn.Dropout(p=opt.dropout)
        self.layer5 = EquivariantLayer(128, opt.classes, activation=None, normalization=None)

    @_head_inference
    def forward(self, x_decentered, x, centers, sn, label, first_pn_out, feature_max_first_pn_out,
                feature_max_knn_feature_1, feature_max_final_pn_out, feature):
        B, N, k = x.size()[0], x.size()[2], self.opt.k
        kN = k * N
        x = torch.cat([x] * k, dim=2)
        sn = torch.cat([sn] * k, dim=2)
        onehot = torch.zeros(B, 16, dtype=torch.float32, device=x.device)
        onehot.scatter_(1, label.unsqueeze(1), 1)
        parts = [x_decentered, x, centers]
        if self.opt.surface_normal:
            parts.append(sn)
        parts += [onehot.unsqueeze(2).expand(B, 16, kN), first_pn_out, feature_max_first_pn_out]
        if self.opt.som_k >= 2:
            parts.append(feature_max_knn_feature_1)
        parts += [feature_max_final_pn_out, feature.unsqueeze(2).expand(B, self.feature_num, kN)]
        h = self.layer3(self.layer2(self.layer1(torch.cat(parts, dim=1))))
        chunks = torch.split(h, self.opt.input_pc_num, dim=2)
        assert len(chunks) == k
        h = chunks[0]
        for c in chunks[1:]:
            h = h + c
        if k > 1:
            h = (1.0 / k) * h if k == 3 else 0.5 * h          # networks.py:331-336 (k in {2, 3})
        h = self.layer4(h)
        if self.opt.dropout > 0.1:
            h = self.drop4(h)
        return self.layer5(h)


    # ---- node-wise evaluation of layer 1 (no-grad / eval) ---------------------------------------------------------
    # Of layer 1's 3356 input channels only 393 vary per point copy (x_decentered, x, sn, first_pn_out); 1923 are
    # node features broadcast back to the copies (centers = the node's coordinates, and the three gathered maps) and
    # 1040 are per-cloud (one-hot label, global feature).  W.[p | n | g] = W_p.x_p + (W_n.x_n + W_g.x_g)[node of the
    # copy]: the node / cloud block is a 64-column GEMM per cloud instead of kN columns (8.5x fewer MACs overall,
    # SURVEY.md section 8f-3).  Column blocks follow the concat order of forward().
    def _layer1_blocks(self):
        sn = 3 if self.opt.surface_normal else 0
        knn = 512 if self.opt.som_k >= 2 else 0
        o = {}
        pos = 0
        for name, width in (("x_dec", 3), ("x", 3), ("centers", 3), ("sn", sn), ("onehot", 16), ("first", 384), ("fm_first", 384),
                            ("fm_knn", knn), ("fm_final", self.feature_num), ("feature", self.feature_num)):
            o[name] = (pos, pos + width)
            pos += width
        return o

    def _layer1_split(self):
        lyr = self.layer1
        w = lyr.conv.weight
        key = (w._version, w.data_ptr(), w.device, _ops.POINTMLP_PRECISION)
        if getattr(self, "_l1_key", None) != key:
            with torch.no_grad():
                W = w.detach().reshape(w.shape[0], w.shape[1]).float()
                blk = self._layer1_blocks()
                cols = lambda names: torch.cat([W[:, blk[n][0]:blk[n][1]] for n in names], dim=1).contiguous()
                mode = _ops.POINTMLP_PRECISION
                self._l1_wp_point = _ops.pointmlp_pack(cols(["first", "x_dec", "x", "sn"]), mode)      # x1 = first_pn_out (384), x2 = small
                self._l1_wp_node = _ops.pointmlp_pack(cols(["centers", "fm_first", "fm_knn", "fm_final"]), mode)
                self._l1_w_glob = cols(["onehot", "feature"])
            self._l1_key = key
        return self._l1_wp_point, self._l1_wp_node, self._l1_w_glob

    def forward_nodewise(self, x_decentered, x, sn, label, first_pn_out, som_node, masked_max, knn_feature_1, final_pn_out, feature,
                         min_idx_i32, first_pn_out_p16=None):
        """Same result as forward() on the gathered tensors, for eval / no-grad: node-level inputs B x C x M + node ids."""
        B, N, k = x.size()[0], x.size()[2], self.opt.k
        lyr = self.layer1
        wp_point, wp_node, w_glob = self._layer1_split()
        scale, shift = lyr._eval_affine()
        Cout = lyr.conv.out_channels
        ones, zeros = _ops.const_vec(Cout, 1.0, x.device), _ops.const_vec(Cout, 0.0, x.device)
        small = [x_decentered, torch.cat([x] * k, dim=2)] + ([torch.cat([sn] * k, dim=2)] if self.opt.surface_normal else [])
        node_in = [som_node, masked_max] + ([knn_feature_1] if self.opt.som_k >= 2 else []) + [final_pn_out]
        onehot = torch.zeros(B, 16, dtype=torch.float32, device=x.device)
        onehot.scatter_(1, label.unsqueeze(1), 1)
        zg = torch.cat([onehot, feature.float()], dim=1) @ w_glob.t()               # B x Cout: per-cloud block
        # the pack decides the kernel and the kernel the storage type: a bf16 pack (precision "bf16") takes bfloat16 operands --
        # the node-level maps arrive as a mix of f32 (som_node, pooled maxima) and bf16 tensors there
        sdt = torch.bfloat16 if wp_point.dtype == torch.int16 else torch.float32
        zn = _ops.pointmlp(torch.cat([t_.to(sdt) for t_ in node_in], dim=1).contiguous(), wp_node, ones, zeros, False, Cout)
        z = (zn.float() + zg.unsqueeze(2)).contiguous()
        x2 = torch.cat(small, dim=1).to(sdt).contiguous()
        c_first = first_pn_out_p16.C if first_pn_out_p16 is not None else first_pn_out.shape[1]
        p16_chain = (wp_point.dtype == torch.int8 and lyr._p16_ok() and self.layer2._p16_ok() and self.layer3._p16_ok() and c_first % 16 == 0)
        if first_pn_out is None:                       # the first PointNet wrote only the P16 planes (segmentation_forward)
            first_pn_out = None if p16_chain else first_pn_out_p16.float()
        if first_pn_out is not None:
            first_pn_out = first_pn_out.to(sdt)
        if p16_chain:
            # third-generation layers: the three kN-column layers hand their activations on pre-split (P16), the split of layer 1's
            # input is one pass over first_pn_out; the per-node block is added in layer 1's epilogue from LDS
            def wsel():
                blk = self._layer1_blocks()
                w = lyr.conv.weight
                W = w.detach().reshape(w.shape[0], w.shape[1]).float()
                return torch.cat([W[:, blk[n][0]:blk[n][1]] for n in ("first", "x_dec", "x", "sn")], dim=1)
            fp16 = first_pn_out_p16 if first_pn_out_p16 is not None else _ops.p16_from_f32(first_pn_out.contiguous())
            h = _ops.pointmlp_h3p(fp16, lyr._packed_p16(wsel, "seg_point"), scale, shift,
                                  lyr.activation == 'relu', Cout, x2=_ops.p16_from_f32(x2), z=z, zidx=min_idx_i32, out="p16")
            h = self.layer2.run_p16(h, out="p16")
            return self._tail(self.layer3.run_p16(h, out="f32"), k)
        if wp_point.dtype == torch.int8 and _ops.POINTMLP_PRECISION == "h3":
            # the per-node block is gathered and added in the per-point launch's epilogue (one pass over B x 1024 x kN less)
            h = _ops.pointmlp_nodeadd(first_pn_out.contiguous(), wp_point, scale, shift, lyr.activation == 'relu', Cout, z, min_idx_i32, x2=x2)
        else:
            t = _ops.pointmlp(first_pn_out.contiguous(), wp_point, ones, zeros, False, Cout, x2=x2)
            h = _ops.node_add_affine_act_(t.float(), z, min_idx_i32, scale, shift, lyr.activation == 'relu')
        return self._tail(self.layer3(self.layer2(h)), k)

    def _tail(self, h, k):
        if (h.is_cuda and h.dtype == torch.float32 and k in (2, 3) and not torch.is_grad_enabled() and h.is_contiguous()
                and h.shape[2] == k * self.opt.input_pc_num and h.shape[0] * h.shape[1] <= 65535):
            h = _ops.chunk_mean(h, k)                          # one launch, the same order of operations (bit-identical)
        else:
            chunks = torch.split(h, self.opt.input_pc_num, dim=2)
            assert len(chunks) == k
            h = chunks[0]
            for c in chunks[1:]:
                h = h + c
            if k > 1:
                h = (1.0 / k) * h if k == 3 else 0.5 * h          # networks.py:331-336 (k in {2, 3})
        h = self.layer4(h)
        if self.opt.dropout > 0.1:
            h = self.drop4(h)
        return self.layer5(h)

    def _nodewise_ok(self):
        lyr = self.layer1
        return (not torch.is_grad_enabled()) and not self.training and lyr.activation in ('relu', None) \
            and lyr.normalization in (None, 'batch') and lyr._fusable() and (_ops.POINTMLP_PRECISION != "h3" or lyr._h3_ok())


def segmentation_forward(encoder, segmenter, pc, sn, label, node, node_knn_I, is_train=False, epoch=None):
    """Model.forward of the part-segmentation task (models/segmenter.py:79-109) on the level-2 encoder:
    the reference recovers the node of every point copy with argmax over the one-hot mask and gathers three
    node-level feature maps back to the kN copies; here the int32 ids are already there and one kernel does
    each gather (autograd falls back to torch.gather when gradients are needed)."""
    encoder.want_first_pn_out = True                  # layer 1 consumes first_pn_out per point copy
    # ... pre-split when the head runs its third-generation chain (the fused first PointNet then writes the P16 planes itself)
    encoder.first_pointnet.emit_p16 = bool(segmenter._nodewise_ok() and getattr(segmenter, "nodewise", True) and segmenter.layer1._p16_ok()
                                           and segmenter.layer2._p16_ok() and segmenter.layer3._p16_ok())
    if encoder.first_pointnet.emit_p16 and _ops.P16_ONLY and _ops.POINTMLP_PRECISION == "h3" and node.size()[2] <= 512:
        # (the pool on the planes -- sonet_index_max_gather_p16 -- holds <= 512 node bins; wider SOMs keep the f32 pool)
        # ... and ONLY pre-split: the per-node pool runs on the planes, first_pn_out is decoded to f32 if somebody reads it (0.3 GB of
        # writes less at 64 x 1024 points)
        encoder.first_pointnet.emit_p16 = "only"
    feature = encoder(pc, sn, node, node_knn_I, is_train, epoch)
    head = lambda: _segmentation_head(encoder, segmenter, pc, sn, label, feature)      # noqa: E731
    if torch.is_grad_enabled() and not segmenter.training and getattr(encoder, "_infer_tag", False):
        with torch.no_grad():                          # inference call (test_model leaves autograd on): same path as under no_grad
            return _ops.run_guarded(head, pc.device, True)
    return _ops.run_guarded(head, pc.device, not (torch.is_grad_enabled() or segmenter.training))


def _segmentation_head(encoder, segmenter, pc, sn, label, feature):
    st = encoder._lazy
    if segmenter._nodewise_ok() and getattr(segmenter, "nodewise", True):
        p16 = getattr(encoder.first_pointnet, "last_p16", None)
        first = None if (p16 is not None and encoder._first_pn_out is None) else encoder.first_pn_out      # (P16-only: not decoded to f32)
        return segmenter.forward_nodewise(encoder.x_decentered, pc, sn, label, first, encoder.som_node,
                                          encoder.first_pn_out_masked_max.contiguous(), encoder.knn_feature_1.contiguous(),
                                          encoder.final_pn_out.contiguous(), feature, st["a"].min_idx_i32,
                                          first_pn_out_p16=p16)
    need_grad = torch.is_grad_enabled() and encoder.first_pn_out_masked_max.requires_grad
    if need_grad:
        idx = encoder.min_idx.unsqueeze(1)
        g1 = torch.gather(encoder.first_pn_out_masked_max, 2, idx.expand(-1, 384, -1))
        g2 = torch.gather(encoder.knn_feature_1, 2, idx.expand(-1, encoder.knn_feature_1.shape[1], -1))
        g3 = torch.gather(encoder.final_pn_out, 2, idx.expand(-1, encoder.final_pn_out.shape[1], -1))
    else:
        ids = st["a"].min_idx_i32
        # (node-level maps are bf16 under precision "bf16": the gather kernel is f32, the maps are B x C x 64)
        g1 = _ops.node_gather(encoder.first_pn_out_masked_max.float().contiguous(), ids)
        g2 = _ops.node_gather(encoder.knn_feature_1.float().contiguous(), ids)
        g3 = _ops.node_gather(encoder.final_pn_out.float().contiguous(), ids)
    return segmenter(encoder.x_decentered, pc, encoder.centers, sn, label, encoder.first_pn_out, g1, g2, g3, feature)


# ---------------------------------------------------------------------------------- autoencoder decoder
class DecoderLinear(nn.Module):
    """FC decoder feature -> 3 x output_fc_pc_num points (models/networks.py:347-369): widths P*2, P*3, P*4, then
    P*3 outputs without norm/activation whose bias starts uniform in [-1, 1].  B x C only: plain PyTorch-ROCm."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.feature_num = opt.feature_num
        self.output_point_number = P = opt.output_fc_pc_num
        widths = [self.feature_num, 2 * P, 3 * P, 4 * P]
        for i in range(3):
            setattr(self, "linear%d" % (i + 1), MyLinear(widths[i], widths[i + 1], activation=opt.activation,
                                                          normalization=opt.normalization))
        self.linear_out = MyLinear(4 * P, 3 * P, activation=None, normalization=None)
        self.linear_out.linear.bias.data.uniform_(-1, 1)

    def forward(self, x):
        for name in ("linear1", "linear2", "linear3", "linear_out"):
            x = getattr(self, name)(x)
        return x.view(-1, 3, self.output_point_number)


class ConvToPC(nn.Module):
    """Two 1x1 convs C -> C -> 3 turning a feature map into a point grid (models/networks.py:372-390)."""

    def __init__(self, in_channels, opt):
        super().__init__()
        self.in_channels = in_channels
        self.opt = opt
        from .layers import MyConv2d
        self.conv1 = MyConv2d(in_channels, int(in_channels), kernel_size=1, stride=1, padding=0, bias=True,
                              activation=opt.activation, normalization=opt.normalization)
        self.conv2 = MyConv2d(int(in_channels), 3, kernel_size=1, stride=1, padding=0, bias=True, activation=None,
                              normalization=None)
        self.conv2.conv.bias.data.uniform_(-1, 1)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class DecoderConv(nn.Module):
    """Up-convolution pyramid 1x1 -> 64x64 with point heads at 16x16, 32x32 and 64x64 (models/networks.py:393-431)."""

    def __init__(self, opt):
        super().__init__()
        from .layers import UpConv
        self.opt = opt
        self.feature_num = F = opt.feature_num
        self.output_point_num = opt.output_conv_pc_num
        chans = [F, int(F), int(F / 2), int(F / 4), int(F / 8), int(F / 8), int(F / 8)]
        for i in range(6):
            setattr(self, "deconv%d" % (i + 1), UpConv(chans[i], chans[i + 1], activation=opt.activation,
                                                         normalization=opt.normalization))
            if i >= 3:
                setattr(self, "conv2pc%d" % (i + 1), ConvToPC(chans[i + 1], opt))

    def forward(self, x):
        x = x.view(-1, self.feature_num, 1, 1)
        for i in range(1, 7):
            x = getattr(self, "deconv%d" % i)(x)
            if i >= 4:
                setattr(self, "pc%d" % i, getattr(self, "conv2pc%d" % i)(x))
        return self.pc6


class Decoder(nn.Module):
    """FC points (if output_fc_pc_num > 0) concatenated with the conv pyramid's 1024- or 4096-point level
    (models/networks.py:434-466); ``conv_pc4`` / ``conv_pc5`` / ``conv_pc6`` / ``linear_pc`` stay readable for the
    multi-resolution Chamfer terms of models/autoencoder.py:89-104."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if opt.output_fc_pc_num > 0:
            self.fc_decoder = DecoderLinear(opt)
        self.conv_decoder = DecoderConv(opt)

    @_head_inference
    def forward(self, x):
        opt = self.opt
        if opt.output_fc_pc_num > 0:
            self.linear_pc = self.fc_decoder(x)
        if opt.output_conv_pc_num > 0:
            self.conv_pc6 = self.conv_decoder(x).view(-1, 3, 4096)
            self.conv_pc4 = self.conv_decoder.pc4.view(-1, 3, 256)
            self.conv_pc5 = self.conv_decoder.pc5.view(-1, 3, 1024)
        conv = {4096: "conv_pc6", 1024: "conv_pc5"}.get(opt.output_conv_pc_num)
        if opt.output_fc_pc_num == 0:
            return getattr(self, conv) if conv else None
        if conv:
            return torch.cat([self.linear_pc, getattr(self, conv)], 2)
        return self.linear_pc
