"""Point-wise layer blocks -- mirror of the reference's models/layers.py on the MI355X kernels.

Class names, constructor signatures, ``forward(x, epoch=None)`` signatures and ``state_dict`` keys
(``conv.weight``, ``conv.bias``, ``norm.weight``, ``norm.running_mean`` ..., ``layers.N.*``,
``linear.*``) are those of the reference, so checkpoints and the unmodified
heads (models/networks.py Classifier / Segmenter / Decoder*) load and run on top.

What is different underneath: ``EquivariantLayer`` (Conv1d k=1 + BN + ReLU, models/layers.py:243-296)
and ``MyConv2d`` with a 1x1 kernel (:169-211) execute as ONE fused gfx950 kernel
(``sonet_pointmlp_f32``: exact-f32 MFMA GEMM with the bias / BatchNorm / ReLU epilogue) instead of
three aten launches; ``PointResNet`` feeds its skip concat (:431) to that kernel as a second input
panel instead of materialising it.  Training mode computes the batch statistics with
``sonet_channel_stats_f32`` and normalises in place; gradients flow through a custom
``autograd.Function``.  ``MyLinear`` / ``UpConv`` (classifier FCs, decoder convs) are outside the
hot path and stay on PyTorch-ROCm ops.

There is no CPU path: calling a fused layer on CPU tensors raises (modules can still be
constructed on CPU, e.g. to build or load a ``state_dict``).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from sonet_hip import ops as _ops
from . import operations

H3_CHECK_EVERY = 16                                              # training: weight-side range test every so many weight versions


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


def _make_act(name):
    if name == 'relu':
        return nn.ReLU()
    if name == 'elu':
        return nn.ELU(alpha=1.0)
    if name == 'swish':
        return Swish()
    if name == 'leakyrelu':
        return nn.LeakyReLU(0.1)
    return None


class _PlainAttrs:
    """Mixin in front of nn.Module: an assignment to an attribute that already lives in the instance ``__dict__`` (intermediate tensors
    kept on the module as the reference keeps them, packed-weight caches and their keys) goes there directly.  nn.Module.__setattr__
    walks the parameter / buffer / module tables first: 5 us per assignment, 40 assignments per training step.  Parameters, modules and
    first assignments take the normal path (a name in ``__dict__`` is never a registered parameter, buffer or module)."""

    def __setattr__(self, name, value):
        d = self.__dict__
        if name in d and not isinstance(value, (nn.Parameter, nn.Module)):
            d[name] = value
        else:
            nn.Module.__setattr__(self, name, value)


class _DecayingBatchNorm(_BatchNorm):
    """BatchNorm whose momentum decays with the epoch (models/layers.py:48-70, :99-120):
    momentum = max(0.01, momentum_original * decay ** (epoch // step)) for epoch >= 1."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, momentum_decay_step=None, momentum_decay=1):
        super().__init__(num_features, eps, momentum, affine)
        self.momentum_decay_step = momentum_decay_step
        self.momentum_decay = momentum_decay
        self.momentum_original = self.momentum

    def decay_momentum(self, epoch):
        step = self.momentum_decay_step
        if epoch is not None and epoch >= 1 and step is not None and step > 0:
            self.momentum = max(0.01, self.momentum_original * (self.momentum_decay ** (epoch // step)))

    def forward(self, input, epoch=None):
        self._check_input_dim(input)
        self.decay_momentum(epoch)
        return F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias,
                            self.training, self.momentum, self.eps)


class MyBatchNorm1d(_DecayingBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() not in (2, 3):
            raise ValueError('expected 2D or 3D input (got {}D input)'.format(input.dim()))


class MyBatchNorm2d(_DecayingBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))


# ---------------------------------------------------------------------------------------------------
# fused conv1x1 + affine + relu
# ---------------------------------------------------------------------------------------------------
def _pack_transposed(weight2d, C1, C2):
    """Packed W[:, :C1]^T (and W[:, C1:]^T) for the dgrad launches: g_x = W^T g_raw is the same fused 1x1-conv
    kernel with the roles of the channel axes swapped."""
    mode = _ops.POINTMLP_PRECISION
    if mode == "h3":
        mode = "x3"                                   # gradients span the f32 exponent range: bf16 pieces, not fp16
    # ("bf16": gradients in bf16 storage too -- same exponent range as f32 -- on one bf16 MFMA per product)
    out = []
    lo = 0
    for Ci in (C1, C2):
        if Ci == 0:
            out.append(None)
            continue
        Cout = weight2d.shape[0]
        Cp = Ci
        if mode != "f32" and Ci % 32 != 0 and Ci > 32:
            # the split-operand kernels want 32-row output tiles: pad with zero rows and drop them afterwards (387 and 515
            # input channels in the KNN module / final PointNet would otherwise fall back to the exact-f32 kernel: 10x slower)
            Cp = (Ci + 31) // 32 * 32
            if (Cp // 32) % 2 == 1 and Cp >= 256:
                # an odd tile count runs one 32-row tile per wave (MT = 1: 2x slower per tile than MT = 4): 13 -> 16, 17 -> 20
                Cp = (Ci + 127) // 128 * 128
        m = mode if (mode == "f32" or _ops.x3_supported(Cout, 0, Cp)) else "f32"
        if (m in ("x3", "h3", "bf16") and weight2d.is_cuda and weight2d.dtype == torch.float32 and weight2d.is_contiguous()
                and (m == "bf16" or Cp % 32 == 0)):
            # packed straight from W (lanes along W's rows, rows Ci .. Cp as zeros): no transposed copy, no padding cat -- the
            # transposed copy + the pack's 32-byte row reads were 0.3 ms of the f32-class training step (six packs per step)
            if _ops.PACK_REGISTRY:
                out.append((_ops.packs.get(weight2d, ("t", m, lo, Ci, Cp)), Ci, Cp))     # (refreshed with every other stale pack in one launch)
            else:
                out.append((_ops.pointmlp_pack_transposed(weight2d.detach(), lo, Ci, Cp, m), Ci, Cp))
        else:
            wt = weight2d[:, lo:lo + Ci].t().contiguous().float()              # Ci x Cout
            if Cp != Ci:
                wt = torch.cat((wt, wt.new_zeros(Cp - Ci, wt.shape[1])), dim=0)
            out.append((_ops.pointmlp_pack(wt, m), Ci, Cp))
        lo += Ci
    return out


def _dgrad(g_raw, pack, acc=None):
    """W^T . g_raw through the fused 1x1-conv kernel (pack from _pack_transposed).  acc (bf16 packs; the caller checked ``_dgrad_acc_ok``):
    another gradient of the same tensor, added by the store of the launch."""
    wpt, Ci, Cp = pack
    want = torch.bfloat16 if wpt.dtype == torch.int16 else torch.float32
    if g_raw.dtype != want:
        g_raw = g_raw.to(want)
    y = _ops.pointmlp(g_raw, wpt, _ops.const_vec(Cp, 1.0, g_raw.device), _ops.const_vec(Cp, 0.0, g_raw.device), False, Cp, acc=acc)
    return y if Cp == Ci else y[:, :Ci]


def _dgrad_acc_ok(g_raw, pack, acc):
    wpt, Ci, Cp = pack
    return (wpt.dtype == torch.int16 and Cp == Ci and acc.dtype == torch.bfloat16 and acc.is_contiguous() and g_raw.shape[2] % 2 == 0
            and tuple(acc.shape) == (g_raw.shape[0], Cp, g_raw.shape[2]))


def _wgrad(g_raw, x, xaff=None):
    """sum over clouds of g_raw[b] . x[b]^T  (Cout x Ci, f32).  xaff = (scale, shift, relu): x holds the RAW output of a BatchNorm layer whose
    normalise pass was never run (f32-class training, ``ops.H3_NORM_ON_LOAD``): the split-operand kernel normalises on load, whatever the shape.  f32-class modes: ``sonet_wgrad_x3_f32`` (both operands split
    into three bf16 pieces on the matrix cores, partial blocks summed in a fixed order); exact-f32 mode and tiny problems: one
    batched hipBLASLt GEMM (K = L is the long axis); bf16 operands accumulate and come out in f32 (a bf16 per-cloud partial
    would cost three of the eight significand bits)."""
    if xaff is not None and x.dtype == torch.bfloat16:
        # bf16 training with normalise-on-load: the streaming weight-gradient kernel normalises its x fragments; other shapes write the
        # activation out first (the values the normalise pass would have stored)
        B_, Co_, L_ = g_raw.shape
        if (g_raw.dtype == torch.bfloat16 and _ops.WGRAD_KERNEL and _ops.wgrad_bf16_xaff_ok(B_, Co_, x.shape[1], L_) and Co_ + x.shape[1] >= 192):
            return _ops.wgrad_bf16(g_raw.contiguous(), x.contiguous(), xaff=xaff)
        x, xaff = _ops.channel_affine_act(x.contiguous(), xaff[0], xaff[1], xaff[2]), None
    if xaff is not None:
        return _ops.wgrad_x3(g_raw.contiguous().float(), x.contiguous(), xaff=xaff)
    if g_raw.dtype == torch.bfloat16:
        if (_ops.WGRAD_KERNEL and g_raw.is_cuda and x.dtype == torch.bfloat16 and g_raw.shape[0] * g_raw.shape[2] <= 8192
                and g_raw.shape[1] * x.shape[1] >= 256 * 128 and g_raw.shape[2] % 8 == 0):
            # node-level layers (64 columns per cloud, wide outputs): sonet_wgrad_bf16 -- one bf16 MFMA per product, f32 partial blocks,
            # fixed-order reduction -- 28 / 35 us against 73 / 79 us for the batched library GEMM + sum at 768x515 / 1024x768
            # (tools/bench_wgrad_bf16.py).
            return _ops.wgrad_bf16(g_raw.contiguous(), x.contiguous())
        if (_ops.WGRAD_KERNEL and g_raw.is_cuda and x.dtype == torch.bfloat16 and g_raw.shape[2] % 8 == 0
                and g_raw.shape[0] * ((g_raw.shape[2] + 63) // 64) >= 2048 and g_raw.shape[1] + x.shape[1] >= 192):
            # point-level layers (15000 columns per cloud): the streaming generation of sonet_wgrad_bf16 (both operands through an LDS-DMA
            # ring, whole 128-byte row segments per request): 95 / 192 us on 128x64 / 256x128 = the library's 96 / 190 (the first
            # generation: 128 / 260); the 64x6 gradient (70 rows in a 256-row ring slot: 80 vs 34 us) stays on hipBLASLt
            return _ops.wgrad_bf16(g_raw.contiguous(), x.contiguous())
        xt = x.transpose(1, 2)
        try:
            return torch.bmm(g_raw, xt, out_dtype=torch.float32).sum(0)
        except (TypeError, RuntimeError):
            return torch.bmm(g_raw.float(), xt.float()).sum(0)
    if (_ops.WGRAD_KERNEL and _ops.POINTMLP_PRECISION in ("h3", "x3") and g_raw.is_cuda and g_raw.dtype == torch.float32
            and x.dtype == torch.float32 and g_raw.shape[0] * g_raw.shape[2] >= 4096 and g_raw.shape[1] * x.shape[1] >= 256 * 128):
        # split-operand MFMA kernel, deterministic; measured against the library GEMM + sum (profiles/r02za_wgrad.log): 256x128 at
        # 64 x 15000 columns 450 vs 533 us, the node-level shapes 55-142 vs 83-182 us; narrow gradients (64x6, 128x64) stay on
        # hipBLASLt (150 vs 50, 195 vs 170 us: too few rows per column slice to keep the loads busy)
        return _ops.wgrad_x3(g_raw.contiguous(), x.contiguous())
    return torch.bmm(g_raw, x.transpose(1, 2)).sum(0)


def _stats_epilogue_ok(x1, wp, Cout):
    """The training forward takes BatchNorm's batch statistics from the layer kernel's epilogue (big tensors, split-operand / bf16 packs)."""
    return (_ops.STATS_EPILOGUE and Cout % 32 == 0 and x1.shape[0] * x1.shape[2] * Cout * 4 >= (32 << 20)
            and ((wp.dtype in (torch.int8, torch.uint8) and x1.dtype == torch.float32)
                 or (wp.dtype == torch.int16 and x1.dtype == torch.bfloat16)))


def _bwd_sums_hint(gy, raw, relu):
    """The BatchNorm-backward sums of the layer whose output gradient is ``gy`` when the launch that produced gy computed them in its
    epilogue (``ops.pointmlp_x3_bnb(..., below=...)``) -- valid only if gy is still exactly that tensor: same object, never added to (autograd
    accumulates a second consumer's gradient IN PLACE: the version counter moves), and computed against this very raw tensor."""
    h = getattr(gy, "_sonet_bwd_sums", None)
    if h is None:
        return None
    sums, raw_ptr, version, hrelu = h
    if raw_ptr != raw.data_ptr() or version != gy._version or hrelu != bool(relu) or sums.numel() != 2 * raw.shape[1] or not gy.is_contiguous():
        return None
    return sums


def _leaf_of(weight2d):
    """The parameter behind a layer's weight2d view (a weak reference: autograd contexts must not keep modules alive)."""
    import weakref
    base = weight2d._base if weight2d._base is not None else weight2d
    try:
        return weakref.ref(base)
    except TypeError:
        return None


class _UseToken:
    """One forward use of a weight by a node whose backward launches its weight gradient on the side stream.  It lives as long as the
    node's context (i.e. the autograd graph) does; ``done`` once that node's backward has run."""
    __slots__ = ("done", "task", "__weakref__")

    def __init__(self):
        self.done = False
        self.task = -1                        # id of the autograd pass (graph task) its backward ran in


def _register_use(wleaf):
    """Called by the forward of such a node (grad mode): remembers on the PARAMETER that one more node of a live graph will produce a
    gradient for it.  Returns the token the node keeps in its context."""
    import weakref
    p = wleaf() if wleaf is not None else None
    if p is None:
        return None
    live = getattr(p, "_sonet_uses", None)
    if live is None:
        live = weakref.WeakSet()
        try:
            p._sonet_uses = live
        except (AttributeError, RuntimeError):
            return None
    tok = _UseToken()
    live.add(tok)
    return tok


def _grad_slot_empty(wleaf, token=None):
    """True when NOTHING reads the weight gradient this backward returns before the backward pass is over -- only then may its side-stream
    launches stay un-joined until the end of the pass:
      * it will be STORED as the parameter's .grad, not added to an existing one (``p.grad is None``);
      * this node is the parameter's ONLY producer in the graphs alive now (a weight used twice -- a Siamese encoder, the encoder called
        twice before one backward -- has its two gradients summed by the engine on the MAIN stream as soon as both exist);
      * the parameter carries no tensor hook and no post-accumulate-grad hook (they run on the main stream inside the pass) -- except
        hooks that declare ``_sonet_joins_side_streams`` (``sonet_hip.dp.GradientAllReducer``'s: it calls ``ops.join_side_streams()`` before it
        touches any gradient: tests/dp_rccl_worker.py checks the step bit for bit against per-layer joins).  The deferral is worth 0.3 ms of
        the 7.75 ms f32-class step and 0.1 ms of the bf16 step (docs/findings.md R6.5).
    Hooks registered on the AccumulateGrad NODE from C++ (torch DistributedDataParallel's reducer) are not visible from here: under DDP
    set SONET_DEFER_WGRAD_JOIN=0 (training section); ``sonet_hip.dp.GradientAllReducer`` joins the side streams itself."""
    p = wleaf() if wleaf is not None else None
    if p is None or p.grad is not None:
        return False
    if getattr(p, "_backward_hooks", None):
        return False
    pah = getattr(p, "_post_accumulate_grad_hooks", None)
    if pah and not all(getattr(h, "_sonet_joins_side_streams", False) for h in pah.values()):
        return False
    if token is None:
        return False
    if token.done:                            # (a second backward through a retained graph: be conservative)
        return False
    task = _ops._graph_task_id() if _ops._graph_task_id is not None else -1
    token.done, token.task = True, task
    live = getattr(p, "_sonet_uses", None)
    if live is None:
        return False
    # Another live use counts when its backward has not run yet OR has run in THIS pass: the engine adds the two gradients on the main stream
    # as soon as the second exists, so neither producer of a shared weight may leave its launch un-joined.  A use whose backward ran in an
    # EARLIER pass does not count: an ordinary training loop keeps the previous step's loss -- and with it that step's graph nodes and their
    # tokens -- alive while the next step runs (``loss, n = step()``), which switched the deferral off for every step but the first.
    # (Outside an autograd pass, or on a torch without the graph-task query, passes cannot be told apart: every other live use counts.)
    return not any((t is not token) and (not t.done or task == -1 or t.task == -1 or t.task == task) for t in live)


class _Materialise(torch.autograd.Function):
    """A deferred activation (the RAW output of a BatchNorm layer standing for act(raw * scale + shift), see ``_PointwiseFn`` ``defer``)
    written out after all: for a consumer without the normalise-on-load form.  The handle's gradient IS the activation's: identity."""

    @staticmethod
    def forward(ctx, raw, sc, sh, relu):
        ctx.set_materialize_grads(False)      # (a consumer that hands its gradient on through a ``_GradCarry`` returns None: no zero tensor for it)
        return _ops.channel_affine_act(raw, sc, sh, relu)

    @staticmethod
    def backward(ctx, gy):
        return gy, None, None, None


class _GradCarry:
    """A gradient handed from one autograd node to another outside autograd's own accumulation.  The first layer's output of the first
    PointNet has two consumers (the second layer and the last one, models/layers.py:417-431); the last layer's backward runs first and its
    input gradient would wait in the engine's buffer until the second layer's arrives, to be added by a pass of its own (three tensors
    through memory).  Instead the early node ``put``s its gradient here and returns None, the late node ``take``s it and has the store of its
    input-gradient launch add it (``ops.pointmlp_x3_bnb(acc=...)``), or adds it itself where that launch does not run.  A deposit is valid
    inside the backward pass that made it only."""
    __slots__ = ("g", "task")

    def __init__(self):
        self.g, self.task = None, -1

    def put(self, g):
        self.g, self.task = g, _ops._graph_task_id()

    def take(self):
        g, task = self.g, self.task
        self.g, self.task = None, -1
        return g if (g is not None and task == _ops._graph_task_id()) else None


class _PointwiseFn(torch.autograd.Function):
    """Differentiable fused layer.  forward: HIP kernels.  backward: two HIP passes turn gy into g_raw (ReLU mask,
    BatchNorm backward: ``sonet_pointwise_bwd_stats/apply``), dgrad = W^T g_raw on the pointmlp kernel, wgrad one
    batched GEMM (hipBLASLt through torch.bmm; K = L is the long axis).

    mode 'affine': y = act((W x) * scale + shift) with constant (scale, shift) (eval BN / no norm);
    mode 'batch' : training BatchNorm -- statistics over (B, L) of raw = W x + bias.
    """

    @staticmethod
    def forward(ctx, x1, x2, weight2d, bias, gamma, beta, wp, scale, shift, relu, mode, eps, bn_run=None, xaff=None, defer=False, carry=None):
        # xaff = (s1, h1, relu1[, s2, h2, relu2]) ('batch' mode, h3 pack): x1 / x2 are RAW outputs of BatchNorm layers standing for
        # act(raw * s + h); the operand loads of this layer and of its weight gradient normalise them.  defer: this layer's own
        # normalise pass is left to ITS consumers -- the first output is raw (standing for the activation: its incoming gradient is the
        # activation's), followed by (mean, var, sc, sh).
        Cout = weight2d.shape[0]
        dev = x1.device
        ones, zeros = _ops.const_vec(Cout, 1.0, dev), _ops.const_vec(Cout, 0.0, dev)
        ctx.xaff = xaff
        ctx.carry = carry                     # (_GradCarry) another consumer of x1 leaves its gradient there: this node returns the sum
        if mode == 'affine':
            y = _ops.pointmlp(x1, wp, scale, shift, relu, Cout, x2=x2)
            # mask source: the output itself (y > 0 <=> pre-activation > 0)
            ctx.save_for_backward(x1, x2 if x2 is not None else x1.new_empty(0), weight2d, scale, y if relu else x1.new_empty(0),
                                  ones, zeros)
        else:
            # bn_run = (running_mean, running_var, momentum, unbias) or None: the statistics launch below also writes the normalisation
            # coefficients and updates the running statistics (``ops.bn_rider``: one launch instead of three per layer and step)
            epi = _stats_epilogue_ok(x1, wp, Cout)
            if xaff is not None and not (epi and wp.dtype in (torch.int8, torch.int16)):
                raise RuntimeError("_PointwiseFn: normalise-on-load needs the h3 or the bf16 layer with the statistics epilogue")
            if not epi:
                raw = _ops.pointmlp(x1, wp, ones, bias, False, Cout, x2=x2)
            rm, rv, mom, unb = bn_run if bn_run is not None else (None, None, 0.0, 1.0)
            invstd, sc, sh = _ops.bn_rider(gamma, beta, eps, rm, rv, mom, unb)
            if epi:
                # batch statistics out of the layer kernel's epilogue (one pass over raw less; big tensors: the small node-level
                # ones keep the second-generation kernel, which has no statistics epilogue)
                raw, mean, var = _ops.pointmlp_stats(x1, wp, ones, bias, False, Cout, x2=x2, xaff=xaff)
            else:
                mean, var = _ops.channel_stats(raw)
            y = raw if defer else _ops.channel_affine_act(raw, sc, sh, relu)
            ctx.save_for_backward(x1, x2 if x2 is not None else x1.new_empty(0), weight2d, sc, sh, raw, mean, invstd, gamma,
                                  zeros)
            if defer:
                ctx.mark_non_differentiable(mean, var, sc, sh)
            else:
                ctx.mark_non_differentiable(mean, var)
        # (without this autograd hands backward freshly zero-filled "gradients" of mean and var: two fill launches per layer and step)
        ctx.set_materialize_grads(False)
        ctx.relu, ctx.mode, ctx.has_x2 = relu, mode, x2 is not None
        ctx.wleaf = _leaf_of(weight2d)
        ctx.use_token = _register_use(ctx.wleaf) if ctx.needs_input_grad[2] else None
        if mode == 'affine':
            return y
        if defer:
            return y, mean, var, sc, sh
        return y, mean, var

    @staticmethod
    def backward(ctx, gy, *unused):
        carried = ctx.carry.take() if ctx.carry is not None else None
        if gy is None:                                                    # (the output was not used)
            return (carried,) + (None,) * 15
        ctx.defer_ok = _grad_slot_empty(ctx.wleaf, ctx.use_token)
        saved = ctx.saved_tensors
        x1, x2, weight2d = saved[:3]
        gy_in = gy
        gy = gy.contiguous()
        g_gamma = g_beta = g_bias = None
        fused_gx1 = None
        if ctx.mode == 'affine':
            sc, y, ones, zeros = saved[3:7]
            src = y if ctx.relu else gy                                   # relu=False: the mask input is unused
            if ctx.needs_input_grad[3]:
                s1, _ = _ops.pointwise_bwd_stats(gy, src, ones, zeros, ctx.relu)
                g_bias = (sc.double() * s1).float()
            if not ctx.relu and sc.data_ptr() == ones.data_ptr():
                g_raw = gy                                                # no activation, unit scale (layers without a norm): identity
            else:
                g_raw = _ops.pointwise_bwd_apply(gy, src, ones, zeros, ctx.relu, sc.contiguous(), zeros, zeros)
        else:
            sc, sh, raw, mean, invstd, gamma, zeros = saved[3:10]
            n = float(raw.shape[0] * raw.shape[2])
            # (the launch that produced gy may have left this layer's sums on it: _bwd_sums_hint)
            sums = _bwd_sums_hint(gy_in, raw, ctx.relu)
            if sums is None:
                sums = _ops.pointwise_bwd_stats(gy, raw, sc, sh, ctx.relu, want_sums=True)
            a, b, c0, g_gamma, g_beta = _ops.bn_bwd_coeffs(sums, mean, invstd, gamma, n)
            g_bias = zeros.clone()                                        # a bias in front of BatchNorm has no gradient
            # f32-class, one input panel, an input gradient to compute: the BatchNorm / ReLU backward rides on the operand load of the
            # dgrad launch (which also writes g_raw for the weight gradient) instead of being a pass of its own over (gy, raw)
            if (_ops.BNB_ON_LOAD and ctx.needs_input_grad[0] and not ctx.has_x2 and gy.dtype == torch.float32 and gy.is_cuda
                    and raw.shape[1] <= 512 and _ops.POINTMLP_PRECISION in ("h3", "x3")):
                pk = _pack_transposed(weight2d, x1.shape[1], 0)[0]
                if pk is not None and pk[0].dtype == torch.uint8:
                    wpt, Ci, Cp = pk
                    dev = gy.device
                    xa_ = ctx.xaff
                    # x1 is the RAW output of the layer below (normalise-on-load) and nothing is padded: this launch's output is that
                    # layer's gy, and its BatchNorm-backward sums come out of the epilogue (handed over on the tensor: _bwd_sums_hint)
                    below = (x1, xa_[0], xa_[1], xa_[2]) if (_ops.BWD_STATS_EPILOGUE and _ops.variants_only() and xa_ is not None and Cp == Ci and x1.dtype == torch.float32) else None
                    acc = None
                    if (carried is not None and below is None and Cp == Ci and carried.dtype == torch.float32 and carried.is_contiguous()
                            and tuple(carried.shape) == (gy.shape[0], Cp, gy.shape[2])):
                        acc, carried = carried, None
                    res = _ops.pointmlp_x3_bnb(gy, raw, wpt, _ops.const_vec(Cp, 1.0, dev), _ops.const_vec(Cp, 0.0, dev), a, b, c0, sc, sh, ctx.relu,
                                               Cp, want_g_raw=ctx.needs_input_grad[2], below=below, acc=acc)
                    yb, g_raw = res[0], res[1]
                    fused_gx1 = yb if Cp == Ci else yb[:, :Ci]
                    if below is not None:
                        fused_gx1._sonet_bwd_sums = (res[2], x1.data_ptr(), fused_gx1._version, bool(xa_[2]))
            # bf16, one input panel: the same on the compiler-scheduled bf16 layer kernel (``sonet_pointmlp_bf16_bnb``: one pass over
            # (gy, raw) instead of the apply pass + the dgrad's read of g_raw; a gradient left in the carry is added by its store)
            if (fused_gx1 is None and _ops.BF16_BNB_ON_LOAD and ctx.needs_input_grad[0] and not ctx.has_x2 and gy.dtype == torch.bfloat16
                    and raw.dtype == torch.bfloat16 and gy.is_cuda and _ops.POINTMLP_PRECISION == "bf16"):
                pk = _pack_transposed(weight2d, x1.shape[1], 0)[0]
                if (pk is not None and pk[0].dtype == torch.int16 and pk[2] == pk[1]
                        and _ops.pointmlp_bf16_bnb_ok(raw.shape[1], pk[2], raw.shape[2])
                        and raw.shape[0] * raw.shape[2] >= 65536):
                    wpt, Ci, Cp = pk
                    dev = gy.device
                    acc = None
                    if (carried is not None and carried.dtype == torch.bfloat16 and carried.is_contiguous()
                            and tuple(carried.shape) == (gy.shape[0], Cp, gy.shape[2])):
                        acc, carried = carried, None
                    fused_gx1, g_raw = _ops.pointmlp_bf16_bnb(gy, raw, wpt, _ops.const_vec(Cp, 1.0, dev), _ops.const_vec(Cp, 0.0, dev), a, b, c0, sc, sh,
                                                              ctx.relu, Cp, want_g_raw=ctx.needs_input_grad[2], acc=acc)
            if fused_gx1 is None:
                g_raw = _ops.pointwise_bwd_apply(gy, raw, sc, sh, ctx.relu, a, b, c0)
        g_w = None
        need1, need2 = ctx.needs_input_grad[0], ctx.has_x2 and ctx.needs_input_grad[1]
        # the weight gradient (HBM-bound) and the input gradient (matrix cores) of a layer are independent: two streams
        ss = _ops.side_stream(g_raw.device) if (ctx.needs_input_grad[2] and (need1 or need2)) else None
        if ctx.needs_input_grad[2]:
            xa = ctx.xaff
            def _gw():
                parts = [_wgrad(g_raw, x1, None if xa is None else xa[:3])]
                if ctx.has_x2:
                    parts.append(_wgrad(g_raw, x2, None if xa is None else xa[3:6]))
                return torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
            if ss is not None:
                with ss:
                    g_w = ss.keep(_gw())
                ss.reads(g_raw, x1, x2 if ctx.has_x2 else None, *([t for t in xa if torch.is_tensor(t)] if xa is not None else []))
            else:
                g_w = _gw()
        g_x1 = g_x2 = None
        if fused_gx1 is not None:
            g_x1 = fused_gx1
        elif need1 or need2:
            Cout = weight2d.shape[0]
            ones_i = None
            packs = _pack_transposed(weight2d, x1.shape[1], x2.shape[1] if ctx.has_x2 else 0)
            outs = []
            for i_, (need, pk) in enumerate(((need1, packs[0]), (need2, packs[1]))):
                if not need or pk is None:
                    outs.append(None)
                    continue
                # (bf16: a gradient the other consumer of x1 left in the carry is added by the store of this launch)
                if i_ == 0 and carried is not None and _dgrad_acc_ok(g_raw, pk, carried):
                    outs.append(_dgrad(g_raw, pk, acc=carried))
                    carried = None
                else:
                    outs.append(_dgrad(g_raw, pk))
            g_x1, g_x2 = outs
        if carried is not None:                                           # (not taken by the launch above)
            g_x1 = carried if g_x1 is None else g_x1 + carried.to(g_x1.dtype)
        if ss is not None:
            # (the weight gradient is not needed before the backward pass is over -- unless it is about to be ADDED to an existing .grad)
            ss.join(defer=ctx.defer_ok)
        return g_x1, g_x2, g_w, g_bias, g_gamma, g_beta, None, None, None, None, None, None, None, None, None, None


class _PooledLastLayerFn(torch.autograd.Function):
    """Last (norm-free, activation-free) layer of the first PointNet + per-node arg-max pool as ONE autograd node
    (models/layers.py:431 + models/networks.py:180-185).  Outputs (first_pn_out, first_pn_out_masked_max).  When only
    the pooled output is used downstream (classifier, autoencoder) the incoming gradient of first_pn_out is None and the
    backward runs the sparse dgrad (``sonet_pooled_dgrad_f32``: C*M 320-vectors per cloud instead of a dense W^T GEMM
    over kN columns); if first_pn_out itself is consumed (segmenter) the dense path runs and the pooled gradient is
    scatter-added into it -- exactly what the reference's gather backward does."""

    @staticmethod
    def forward(ctx, x1, x2, weight2d, bias, wp, min_idx_i32, row_max, M, need_dense=True, pos0=None, xaff=None, carry=None):
        Cout = weight2d.shape[0]
        ctx.carry = carry                     # (_GradCarry) x1's other consumer adds this node's x1 gradient to its own
        ones = _ops.const_vec(Cout, 1.0, x1.device)
        b = bias.detach().float().contiguous()
        ctx.pos0 = None
        ctx.wleaf = _leaf_of(weight2d)
        ctx.use_token = _register_use(ctx.wleaf) if ctx.needs_input_grad[2] else None
        ctx.xaff = xaff                       # (sorted form only) x1 / x2 are RAW outputs of BatchNorm layers: see _PointwiseFn
        if xaff is not None and pos0 is None and not (wp.dtype == torch.int16 and not need_dense and _ops.pointmlp_bf16_pool_ok(x1, x2, Cout, M)):
            raise RuntimeError("_PooledLastLayerFn: normalise-on-load comes with the node-sorted form (f32-class) or the bf16 pool epilogue")
        if pos0 is not None:
            # NODE-SORTED columns (the caller ran the hidden layers on som_sort_group's copy; min_idx_i32 = its ids_sorted): the f32-class
            # layer pools its own output -- neither the B x 384 x kN tensor nor the index_max launch exists.  Positions are sorted
            # columns (what x1 / x2 are indexed with in the backward); "position 0" of the reference is the sorted position pos0[b]
            # of original column 0, already in place for bins nothing beat and for empty nodes.
            if wp.dtype == torch.int8 and _ops.pointmlp_h3_segpool_ok(x1, x2, wp, Cout, M):
                idx, val = _ops.pointmlp_h3_segpool(x1, wp, ones, b, False, Cout, min_idx_i32, pos0, M, row_max, x2=x2, xaff=xaff)
            else:
                # (the weight side of the range guard sent the layer to x3 between the caller's check and here: store + index_max on the
                #  sorted tensor, "position 0" moved to pos0 by hand)
                if xaff is not None:
                    raise RuntimeError("_PooledLastLayerFn: normalise-on-load needs the h3 pack")
                y = _ops.pointmlp(x1, wp, ones, b, False, Cout, x2=x2)
                idx0, _ = _ops.index_max_gather(y, min_idx_i32, M, None)
                Bq = x1.shape[0]
                at0 = (min_idx_i32[:, :1] == torch.arange(M, device=x1.device, dtype=torch.int32).view(1, M)).unsqueeze(1) & (y[:, :, :1] > -1000.0)
                never = ((idx0 == 0) & ~at0) | (row_max.unsqueeze(1) == 0)
                idx = torch.where(never, pos0.view(Bq, 1, 1).expand_as(idx0), idx0).contiguous()
                val = torch.gather(y, 2, idx.long())
                del y
            ctx.save_for_backward(x1, x2, weight2d, idx, row_max)
            ctx.pos0 = pos0
            ctx.set_materialize_grads(False)
            ctx.mark_non_differentiable(idx)
            return None, val, idx
        if not need_dense and wp.dtype == torch.int16 and _ops.pointmlp_bf16_pool_ok(x1, x2, Cout, M):
            # nobody reads first_pn_out itself (classifier, autoencoder): the layer's epilogue IS the pool -- the B x 384 x kN tensor is
            # neither written nor read back (0.74 GB of HBM traffic each way at B = 64) and the index_max launch is gone; positions and
            # values are those of index_max_gather on the tensor the storing launch would have written, bit for bit
            y = None
            idx, val = _ops.pointmlp_bf16_pool(x1, wp, ones, b, False, Cout, min_idx_i32, M, row_max, x2=x2, xaff=xaff)
        else:
            y = _ops.pointmlp(x1, wp, ones, b, False, Cout, x2=x2)
            idx, val = _ops.index_max_gather(y, min_idx_i32, M, row_max)
        gi = idx * row_max.unsqueeze(1)                                   # networks.py:185: empty nodes gather position 0
        ctx.save_for_backward(x1, x2, weight2d, gi, row_max)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(gi)
        return y, val, gi

    @staticmethod
    def backward(ctx, g_y, g_mm, _g_gi):
        x1, x2, weight2d, gi, row_max = ctx.saved_tensors
        B, C1, L = x1.shape
        C2 = x2.shape[1]
        if g_y is None and g_mm is None:
            return (None,) * 12
        sparse = g_y is None
        xa = ctx.xaff
        xa1, xa2 = (None, None) if xa is None else (xa[:3], xa[3:6])
        G = None
        ss = None
        if g_mm is not None:
            g_mm = g_mm.contiguous()
        if not sparse:
            G = g_y.contiguous().clone() if g_mm is not None else g_y.contiguous()
            if g_mm is not None:
                G.scatter_add_(2, gi.long(), g_mm.to(G.dtype))            # the gather's backward (duplicates accumulate)
        g_bias = None
        if ctx.needs_input_grad[3]:
            # sparse: the gradient of first_pn_out is the scatter of g_mm and nothing else (never built): its sum is the sum of g_mm
            g_bias = g_mm.float().sum(dim=(0, 2)) if sparse else G.float().sum(dim=(0, 2))
        g_w = None
        if ctx.needs_input_grad[2]:
            if g_mm is not None and g_mm.dtype != torch.float32:
                g_mm = g_mm.float()
            if sparse and g_mm.shape[1] <= 384 and g_mm.shape[2] <= 64 and L * 8 <= 152 * 1024:
                # 24,576 entries per cloud instead of a dense GEMM over kN columns (the kernel's limits: C, M, two rows in LDS);
                # on the side stream: the sparse dgrad below does not depend on it (0.66 + 0.83 ms in sequence otherwise)
                ss = _ops.side_stream(x1.device) if ((ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and _ops.POOLED_SIDE_STREAM) else None
                if ss is not None:
                    with ss:
                        g_t, gi_t = g_mm.transpose(1, 2).contiguous(), gi.transpose(1, 2).contiguous()
                        g_w = ss.keep(torch.cat((_ops.pooled_wgrad(g_t, gi_t, x1, xa1), _ops.pooled_wgrad(g_t, gi_t, x2, xa2)), dim=1))
                    ss.reads(g_mm, gi, x1, x2, *([t for t in xa if torch.is_tensor(t)] if xa is not None else []))
                else:
                    g_t, gi_t = g_mm.transpose(1, 2).contiguous(), gi.transpose(1, 2).contiguous()     # B x M x C: coalesced entry loads
                    g_w = torch.cat((_ops.pooled_wgrad(g_t, gi_t, x1, xa1), _ops.pooled_wgrad(g_t, gi_t, x2, xa2)), dim=1)
            else:
                if G is None:
                    G = torch.zeros((B, weight2d.shape[0], L), dtype=x1.dtype, device=x1.device)
                    G.scatter_add_(2, gi.long(), g_mm.to(x1.dtype))
                g_w = torch.cat((_wgrad(G, x1, xa1), _wgrad(G, x2, xa2)), dim=1)
        g_x1 = g_x2 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if sparse:
                # every channel of an EMPTY node gathers position 0: those C entries per node are one dense mat-vec into
                # column 0 (left in the sparse kernel they would pile thousands of entries onto a single tile)
                w = weight2d.detach().float().contiguous()
                occ = row_max.unsqueeze(1) > 0
                wt_pack = None
                if x1.dtype == torch.bfloat16 and _ops.pooled_dgrad_mfma_ok(w.shape[0], C1, C2, L):
                    Cp = (C1 + C2 + 31) // 32 * 32                                 # (C1 + C2) x C, rows padded to 32-row tiles
                    if _ops.PACK_REGISTRY and weight2d.is_contiguous() and weight2d.dtype == torch.float32:
                        wt_pack = _ops.packs.get(weight2d, ("t", "bf16", 0, C1 + C2, Cp))
                    else:
                        wt = w.t().contiguous()
                        if wt.shape[0] % 32:
                            wt = torch.cat((wt, wt.new_zeros(32 - wt.shape[0] % 32, wt.shape[1])), dim=0)
                        wt_pack = _ops.pointmlp_pack(wt, "bf16")
                col0 = torch.matmul((g_mm.float() * (~occ)).sum(dim=2), w)        # B x (C1 + C2)
                if ctx.pos0 is not None and wt_pack is None and _ops.pooled_dgrad_tail_ok(C1, C2, x1.dtype):
                    # node-sorted f32-class path: the column-0 part is added by the store of the launch, and when x2 IS the raw output of the
                    # layer below (normalise-on-load) that layer's BatchNorm-backward sums come out of the same store, handed over on the
                    # gradient tensor (_bwd_sums_hint): no statistics pass over (gy, raw) of the widest hidden layer
                    below = (x2, xa[3], xa[4], xa[5]) if (xa is not None and C2 > 0 and x2.dtype == torch.float32 and x2.is_contiguous()) else None
                    res = _ops.pooled_dgrad(g_mm.float(), torch.where(occ, gi, torch.full_like(gi, -1)), w, C1, C2, L, col0=col0.contiguous(),
                                            pos0=ctx.pos0.to(torch.int32).contiguous(), below=below)
                    g_x1, g_x2 = res[0], res[1]
                    if below is not None:
                        g_x2._sonet_bwd_sums = (res[2], x2.data_ptr(), g_x2._version, bool(xa[5]))
                    if ss is not None:
                        ss.join(defer=_grad_slot_empty(ctx.wleaf, ctx.use_token))
                    if ctx.carry is not None and g_x1 is not None:
                        ctx.carry.put(g_x1)
                        g_x1 = None
                    return g_x1, g_x2, g_w, g_bias, None, None, None, None, None, None, None, None
                g_x1, g_x2 = _ops.pooled_dgrad(g_mm.float(), torch.where(occ, gi, torch.full_like(gi, -1)), w, C1, C2, L, out_dtype=x1.dtype,
                                               wt_pack=wt_pack)
                if ctx.pos0 is None:
                    g_x1[:, :, 0] += col0[:, :C1].to(g_x1.dtype)
                    g_x2[:, :, 0] += col0[:, C1:].to(g_x2.dtype)
                else:                                                             # node-sorted columns: original column 0 sits at pos0[b]
                    p0 = ctx.pos0.long().view(B, 1, 1)
                    g_x1.scatter_add_(2, p0.expand(B, C1, 1), col0[:, :C1].to(g_x1.dtype).unsqueeze(2))
                    g_x2.scatter_add_(2, p0.expand(B, C2, 1), col0[:, C1:].to(g_x2.dtype).unsqueeze(2))
            else:
                packs = _pack_transposed(weight2d, C1, C2)
                outs = []
                for pk in packs:
                    outs.append(_dgrad(G, pk))
                g_x1, g_x2 = outs
        if ss is not None:
            ss.join(defer=_grad_slot_empty(ctx.wleaf, ctx.use_token))
        if ctx.carry is not None and g_x1 is not None:
            ctx.carry.put(g_x1)
            g_x1 = None
        return g_x1, g_x2, g_w, g_bias, None, None, None, None, None, None, None, None


class _FusedPointwise(_PlainAttrs, nn.Module):
    """Shared implementation of EquivariantLayer / MyConv2d(1x1): holds ``conv`` (+ ``norm``, ``act``)
    exactly like the reference modules and runs them as one kernel."""

    def _fusable(self):
        ks = self.conv.kernel_size
        return all(k == 1 for k in ks) and all(s == 1 for s in self.conv.stride) and all(p == 0 for p in self.conv.padding) \
            and self.conv.groups == 1

    def _weight2d(self):
        w = self.conv.weight
        return w.reshape(w.shape[0], w.shape[1])

    def _h3_ok(self):
        """Weight side of the h3 range guard (sonet_hip.ops.h3_weight_ok), cached per weight version."""
        w = self.conv.weight
        key = (w._version, w.data_ptr(), w.device)
        if getattr(self, '_h3ok_key', None) == key:
            return self._h3ok
        if getattr(self, '_h3ok_dev', None) != w.device:            # moved to another device: a pending verdict belongs to the old copy
            self._h3ok_pending, self._h3ok_dev = None, w.device
            self.__dict__.pop('_h3ok_checked', None)
        if not w.is_cuda:
            self._h3ok, self._h3ok_key = True, key
            return True
        if torch.is_grad_enabled() and w.requires_grad and hasattr(self, '_h3ok'):
            # training: the weight changes every step and a host read would drain the stream once per layer and step.  The test runs
            # on the device every H3_CHECK_EVERY versions and its verdict is read when it has arrived (pinned memory + event) --
            # like the activation side of the guard, a violation takes effect one check late.
            pend = getattr(self, '_h3ok_pending', None)
            if pend is not None and pend[1].query():
                self._h3ok = bool(int(pend[0][0]))
                self._h3ok_pending = pend = None
                if not self._h3ok:
                    _ops.h3_ratio_warn("a point-wise layer's weight (%d x %d)" % (w.shape[0], w.shape[1]))
            # (elapsed versions, not a modulo: an optimizer that bumps the version twice per step would never hit an even multiple)
            if pend is None and w._version - getattr(self, '_h3ok_checked', -H3_CHECK_EVERY) >= H3_CHECK_EVERY:
                self._h3ok_checked = w._version
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(_ops.h3_weight_ratio_flag(self._weight2d()), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._h3ok_pending = (host, ev)
            return self._h3ok
        self._h3ok = _ops.h3_weight_ok(self._weight2d())
        self._h3ok_key = key
        if not self._h3ok:
            _ops.h3_ratio_warn("a point-wise layer's weight (%d x %d)" % (w.shape[0], w.shape[1]))
        return self._h3ok

    def _packed(self, C1=None, C2=0):
        """Packed weight for the current arithmetic mode (``sonet_hip.ops.POINTMLP_PRECISION``)."""
        w = self.conv.weight
        mode = _ops.POINTMLP_PRECISION
        if mode == "h3" and not self._h3_ok():
            mode = "x3"
        if mode in ("x3", "h3", "bf16") and not _ops.x3_supported(w.shape[1] if C1 is None else C1, C2, w.shape[0]):
            mode = "f32"
        if _ops.PACK_REGISTRY and mode != "f32" and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous():
            # one launch refreshes every stale pack of the step (forward and transposed, all layers): sonet_hip.ops.packs
            return _ops.packs.get(self._weight2d(), ("fwd", mode))
        key = (w._version, w.data_ptr(), w.device, mode)
        if getattr(self, '_wp_key', None) != key:
            with torch.no_grad():
                self._wp = _ops.pointmlp_pack(self._weight2d().detach().contiguous().float(), mode)
            self._wp_key = key
        return self._wp

    def _packed_rotated(self, lead):
        """h3 (or bf16) pack of the weight with its first ``lead`` input channels moved behind the others: a caller that feeds
        cat(a, b) with a narrow ``a`` (3 coordinate rows) passes (b, a) as (x1, x2) instead, so that x1 is 16-aligned."""
        w = self.conv.weight
        mode = "bf16" if _ops.POINTMLP_PRECISION == "bf16" else "h3"
        key = (w._version, w.data_ptr(), w.device, lead, mode)
        if getattr(self, '_wpr_key', None) != key:
            with torch.no_grad():
                w2 = self._weight2d().detach().float()
                self._wpr = _ops.pointmlp_pack(torch.cat((w2[:, lead:], w2[:, :lead]), dim=1).contiguous(), mode)
            self._wpr_key = key
        return self._wpr

    def _packed_split(self, lead):
        """(h3 or bf16 pack of W[:, lead:], W[:, :lead] as a contiguous Cout x lead f32 matrix): the layer is linear, so a caller
        whose input is cat(a, gather(b)) applies the wide block to b ONCE per node and adds the narrow block per column."""
        w = self.conv.weight
        mode = "bf16" if _ops.POINTMLP_PRECISION == "bf16" else "h3"
        key = (w._version, w.data_ptr(), w.device, lead, mode)
        if getattr(self, '_wps_key', None) != key:
            with torch.no_grad():
                w2 = self._weight2d().detach().float()
                self._wps = (_ops.pointmlp_pack(w2[:, lead:].contiguous(), mode), w2[:, :lead].contiguous())
            self._wps_key = key
        return self._wps

    def _direct_ok(self, x):
        """Eval-mode, no-grad, h3 / bf16: the caller may launch the kernel itself with this layer's pack and folded affine."""
        return (_ops.GATHER_NODE_STAGE and _ops.POINTMLP_PRECISION in ("h3", "bf16") and not torch.is_grad_enabled() and x.is_cuda
                and (_ops.POINTMLP_PRECISION != "h3" or self._h3_ok())
                and x.dtype in (torch.float32, torch.bfloat16) and self._fusable() and self.conv.out_channels % 32 == 0
                and self.normalization in (None, 'batch') and not (self.normalization == 'batch' and self.norm.training)
                and self.activation in (None, 'relu'))

    def _packed_p16(self, make=None, tag=None):
        """h3p pack (third-generation layer, P16 operands) of this layer's weight -- or of ``make()``, a column selection of it the
        caller keeps apart under ``tag`` (the segmenter's per-point block of layer 1; built only when the weight has changed)."""
        w = self.conv.weight
        key = (w._version, w.data_ptr(), w.device, tag)
        cache = self.__dict__.setdefault('_wp16', {})
        if cache.get(tag, (None, None))[0] != key:
            with torch.no_grad():
                src = self._weight2d().detach() if make is None else make()
                cache[tag] = (key, _ops.pointmlp_h3p_pack(src.contiguous().float()))
        return cache[tag][1]

    def _lead_cols(self, lead):
        """W[:, :lead] as a contiguous Cout x lead f32 matrix (the coordinate channels of KNNModule's first layer), cached per weight version."""
        w = self.conv.weight
        key = (w._version, w.data_ptr(), w.device, lead)
        if getattr(self, '_wlead_key', None) != key:
            with torch.no_grad():
                self._wlead = self._weight2d().detach()[:, :lead].float().contiguous()
            self._wlead_key = key
        return self._wlead

    def _p16_ok(self):
        """Eval-mode, no-grad, h3: this layer can run on pre-split (P16) operands and hand its output on in that form."""
        return (_ops.POINTMLP_PRECISION == "h3" and _ops.P16_CHAINS and not torch.is_grad_enabled() and self.conv.weight.is_cuda
                and self._h3_ok() and self._fusable() and self.conv.out_channels % 32 == 0
                and self.normalization in (None, 'batch') and not (self.normalization == 'batch' and self.norm.training)
                and self.activation in (None, 'relu'))

    def run_p16(self, x, x2=None, out="p16", **kw):
        """The layer on P16 operands (``sonet_hip.ops.P16``; f32 tensors are converted first): out "p16" | "f32" | "both".
        Only when ``_p16_ok()``."""
        if not isinstance(x, _ops.P16):
            x = _ops.p16_from_f32(self._prep(x).float())
        if x2 is not None and not isinstance(x2, _ops.P16):
            x2 = _ops.p16_from_f32(self._prep(x2).float())
        scale, shift = self._eval_affine()
        return _ops.pointmlp_h3p(x, self._packed_p16(), scale, shift, self.activation == 'relu', self.conv.out_channels, x2=x2, out=out, **kw)

    def _bias(self):
        if self.conv.bias is not None:
            return self.conv.bias
        return torch.zeros(self.conv.out_channels, dtype=torch.float32, device=self.conv.weight.device)

    def _eval_affine(self):
        """(scale, shift) folding the conv bias and the eval-mode BatchNorm (running statistics)."""
        bn = self.norm if self.normalization == 'batch' else None
        ts = [self.conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        key = tuple((t._version, t.data_ptr()) if t is not None else None for t in ts) + (self.conv.weight.device,)
        if getattr(self, '_affine_key', None) != key:
            with torch.no_grad():
                b = self._bias().detach().float()
                if bn is None:
                    scale, shift = _ops.const_vec(b.numel(), 1.0, b.device), b.clone()      # the shared ones vector marks a unit scale
                else:
                    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
                    shift = (b - bn.running_mean) * scale + bn.bias.detach()
                self._affine = (scale.contiguous(), shift.contiguous())
            self._affine_key = key
            self._affine_gen = getattr(self, '_affine_gen', 0) + 1          # (id() of the tensors can be recycled: count instead)
        return self._affine

    def _run(self, x1, x2, epoch, xaff=None, defer=False, carry=None):
        """x1 (and optional x2): B x C x L contiguous f32 CUDA tensors -> B x Cout x L.
        Training BatchNorm layers of the f32-class first PointNet (``PointResNet.forward_pooled``): ``xaff`` = (s1, h1, relu1[, s2, h2,
        relu2]) says x1 / x2 are RAW outputs of BatchNorm layers standing for act(raw * s + h) (normalised by this layer's operand loads,
        or written out first when this layer has no such form); ``defer`` -> ((raw, sc, sh), act_done): this layer's own normalise pass is
        left to its consumers."""
        relu_fused = self.activation == 'relu'
        norm = self.normalization
        bn = self.norm if norm == 'batch' else None
        if bn is not None:
            bn.decay_momentum(epoch)
        train_bn = bn is not None and bn.training
        needs_grad = torch.is_grad_enabled() and (x1.requires_grad or (x2 is not None and x2.requires_grad)
                                                  or self.conv.weight.requires_grad)
        fuse_act = relu_fused and (norm in (None, 'batch'))
        wp = self._packed(x1.shape[1], x2.shape[1] if x2 is not None else 0)
        want = torch.bfloat16 if wp.dtype == torch.int16 else torch.float32     # the pack decides the kernel, the kernel the storage type
        if x1.dtype != want:
            x1 = x1.to(want)
        if x2 is not None and x2.dtype != want:
            x2 = x2.to(want)
        if (xaff is not None or defer or carry is not None) and not (train_bn and fuse_act):
            raise RuntimeError("deferred normalisation is a training-mode BatchNorm + ReLU layer's")
        C2_ = x2.shape[1] if x2 is not None else 0
        have_form = xaff is not None and _stats_epilogue_ok(x1, wp, self.conv.out_channels) and (
            (wp.dtype == torch.int8 and _ops.xaff_ok(x1.shape[1], C2_, self.conv.out_channels))
            or (wp.dtype == torch.int16 and _ops.bf16_xaff_ok(x1.shape[0], x1.shape[1], C2_, self.conv.out_channels, x1.shape[2])))
        if xaff is not None and not have_form:
            # (the weight side of the range guard sent this layer to x3 between the caller's check and here, or a shape without the
            #  normalise-on-load form: its inputs are written out after all)
            x1 = _Materialise.apply(x1, xaff[0], xaff[1], xaff[2])
            if x2 is not None:
                x2 = _Materialise.apply(x2, xaff[3], xaff[4], xaff[5])
            xaff = None
        if train_bn:
            n = x1.shape[0] * x1.shape[2]
            m = bn.momentum
            # F.batch_norm's running-statistics update rides on the statistics launch of the forward (``ops.bn_rider``) when the buffers allow
            ride = (bn.track_running_stats and m is not None and bn.running_mean is not None and bn.running_mean.is_contiguous()
                    and bn.running_var.is_contiguous() and bn.running_mean.dtype == torch.float32 and bn.running_mean.device == x1.device)
            outs = _PointwiseFn.apply(x1, x2, self._weight2d(), self._bias(), bn.weight, bn.bias, wp, None, None,
                                      fuse_act, 'batch', bn.eps, (bn.running_mean, bn.running_var, m, n / max(n - 1, 1)) if ride else None,
                                      xaff, bool(defer), carry)
            y, mean, var = outs[:3]
            if defer:
                y = (y, outs[3], outs[4])
            with torch.no_grad():                                       # F.batch_norm running-stat update
                if ride:
                    pass
                elif bn.running_mean is not None and m is not None:
                    bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
                # the kernel writes the buffers through raw pointers: their _version does not move, so the folded eval
                # affine (keyed on versions) must be dropped by hand -- or a later eval() would run on stale statistics
                self._affine_key = None
                # (F.batch_norm does not touch num_batches_tracked; the reference never increments it)
        else:
            if norm in (None, 'batch'):
                scale, shift = self._eval_affine()
            else:                                                        # instance norm etc.: conv only
                b = self._bias().detach().float()
                scale, shift = torch.ones_like(b), b.contiguous()
            if needs_grad:
                y = _PointwiseFn.apply(x1, x2, self._weight2d(), self._bias(), None, None, wp, scale, shift, fuse_act,
                                       'affine', 0.0)
            else:
                y = _ops.pointmlp(x1, wp, scale, shift, fuse_act, self.conv.out_channels, x2=x2)
        return y, fuse_act

    @staticmethod
    def _prep(x):
        if not x.is_cuda:
            raise _ops.SonetHipError("fused point-wise layers run on the MI355X only (got a CPU tensor); "
                                     "there is no CPU fallback")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        return x.contiguous()


class EquivariantLayer(_FusedPointwise):
    def __init__(self, num_in_channels, num_out_channels, activation='relu', normalization=None, momentum=0.1,
                 bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self.num_in_channels = num_in_channels
        self.num_out_channels = num_out_channels
        self.activation = activation
        self.normalization = normalization
        self.conv = nn.Conv1d(num_in_channels, num_out_channels, kernel_size=1, stride=1, padding=0)
        if normalization == 'batch':
            self.norm = MyBatchNorm1d(num_out_channels, momentum=momentum, affine=True,
                                      momentum_decay_step=bn_momentum_decay_step, momentum_decay=bn_momentum_decay)
        elif normalization == 'instance':
            self.norm = nn.InstanceNorm1d(num_out_channels, momentum=momentum, affine=True)
        act = _make_act(activation)
        if act is not None:
            self.act = act
        self.weight_init()

    def weight_init(self):
        n = self.conv.kernel_size[0] * self.conv.in_channels
        self.conv.weight.data.normal_(0, math.sqrt(2. / n))             # models/layers.py:271-275
        if self.conv.bias is not None:
            self.conv.bias.data.fill_(0)
        if self.normalization in ('batch', 'instance'):
            self.norm.weight.data.fill_(1)
            self.norm.bias.data.zero_()

    def forward(self, x, epoch=None, x_skip=None):
        """x: B x Cin x L.  ``x_skip`` (optional, B x C_skip x L, C_skip % 8 == 0) goes first in the channel
        order: the layer computes conv(cat(x_skip, x)) without materialising the concat."""
        if x_skip is not None:
            y, act_done = self._run(self._prep(x_skip), self._prep(x), epoch)
        else:
            y, act_done = self._run(self._prep(x), None, epoch)
        if self.normalization is not None and self.normalization != 'batch':
            y = self.norm(y)
        if self.activation is not None and not act_done:
            y = self.act(y)
        return y


class _FcFn(torch.autograd.Function):
    """Linear + BatchNorm1d (batch statistics) + ReLU of a B x C head layer in training as one forward and two backward launches
    (``sonet_fc_bn_act_fwd_f32`` / ``_bwd_f32`` / ``sonet_fc_dx_f32``; models/layers.py:123-166).  ``bn`` is the BatchNorm module (its
    running statistics are updated in place by the forward kernel, as F.batch_norm does) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, relu):
        x, weight = x.contiguous(), weight.contiguous()
        if bn is not None:
            if x.shape[0] < 2:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
            rm, rv = bn.running_mean, bn.running_var
            if not all(t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == x.device) for t in (rm, rv)):
                raise _ops.SonetHipError("MyLinear: the BatchNorm running statistics must be contiguous float32 tensors on the input's device")
            y, xhat, invstd = _ops.fc_bn_act_fwd(x, weight, bias, gamma.contiguous(), beta.contiguous(), rm, rv, bn.momentum, bn.eps, relu)
            # the kernel wrote the running statistics through raw pointers: move their version counters as an in-place aten op would
            # (the folded eval-mode affine of MyLinear._eval_affine is keyed on them)
            for t in (rm, rv):
                if t is not None:
                    torch.autograd.graph.increment_version(t)
        else:
            y, xhat, invstd = _ops.fc_bn_act_fwd(x, weight, bias, None, None, None, None, 0.0, 0.0, relu)
        ctx.relu, ctx.has_bn, ctx.has_bias = bool(relu), bn is not None, bias is not None
        empty = x.new_empty(0)
        ctx.save_for_backward(x, weight, y if relu else empty, xhat if xhat is not None else empty, invstd if invstd is not None else empty,
                              gamma.detach() if gamma is not None else empty)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y, xhat, invstd, gamma = ctx.saved_tensors
        bn = ctx.has_bn
        dz, dW, db, dgamma, dbeta = _ops.fc_bn_act_bwd(gy.contiguous(), y if ctx.relu else None, xhat if bn else None, invstd if bn else None,
                                                      gamma if bn else None, x, ctx.relu, want_dw=ctx.needs_input_grad[1])
        dx = _ops.fc_dx(dz, weight) if ctx.needs_input_grad[0] else None
        return dx, dW, (db if ctx.has_bias and ctx.needs_input_grad[2] else None), dgamma, dbeta, None, None


class MyLinear(_PlainAttrs, nn.Module):
    """FC + BN + act (models/layers.py:123-166).  Classifier-head layer, B x C only: eval / no-grad on ``sonet_linear_act_f32``,
    training on the ``sonet_fc_*`` kernels (``_FcFn``); other norms / activations / wide batches stay on aten."""

    def __init__(self, in_features, out_features, activation=None, normalization=None, momentum=0.1,
                 bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self.activation = activation
        self.normalization = normalization
        self.linear = nn.Linear(in_features, out_features, bias=True)
        if normalization == 'batch':
            self.norm = MyBatchNorm1d(out_features, momentum=momentum, affine=True,
                                      momentum_decay_step=bn_momentum_decay_step, momentum_decay=bn_momentum_decay)
        elif normalization == 'instance':
            self.norm = nn.InstanceNorm1d(out_features, momentum=momentum, affine=True)
        act = _make_act(activation)
        if act is not None:
            self.act = act
        self.weight_init()

    def weight_init(self):
        self.linear.weight.data.normal_(0, math.sqrt(2. / self.linear.in_features))
        if self.linear.bias is not None:
            self.linear.bias.data.fill_(0)
        if self.normalization in ('batch', 'instance'):
            self.norm.weight.data.fill_(1)
            self.norm.bias.data.zero_()

    def _eval_affine(self):
        """(scale, shift) folding the bias and the eval-mode BatchNorm1d into the matrix product."""
        bn = self.norm if self.normalization == 'batch' else None
        ts = [self.linear.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        key = tuple((t._version, t.data_ptr()) if t is not None else None for t in ts)
        if getattr(self, '_affine_key', None) != key:
            with torch.no_grad():
                b = self.linear.bias.detach().float() if self.linear.bias is not None else torch.zeros(self.linear.out_features, device=self.linear.weight.device)
                if bn is None:
                    scale, shift = torch.ones_like(b), b.clone()
                else:
                    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
                    shift = (b - bn.running_mean) * scale + bn.bias.detach()
                self._affine = (scale.contiguous(), shift.contiguous())
            self._affine_key = key
        return self._affine

    def forward(self, x, epoch=None):
        fast = (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
                and self.normalization in (None, 'batch') and self.activation in (None, 'relu')
                and not (self.normalization == 'batch' and self.norm.training))
        if fast:                                         # eval / no-grad: Linear + BN + ReLU as one kernel
            scale, shift = self._eval_affine()
            return _ops.linear_act(x.contiguous(), self.linear.weight.detach().contiguous(), scale, shift, self.activation == 'relu')
        bn = self.norm if self.normalization == 'batch' else None
        if (torch.is_grad_enabled() and self.normalization in (None, 'batch') and self.activation in (None, 'relu')
                and (bn is None or (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None))
                and _ops.fc_head_ok(x, self.linear.weight) and (bn is None or x.shape[0] >= 2)):
            # training: Linear + BatchNorm1d (batch statistics) + ReLU in one launch, two for the backward
            if bn is not None:
                bn.decay_momentum(epoch)
            return _FcFn.apply(x, self.linear.weight, self.linear.bias, bn.weight if bn is not None else None,
                               bn.bias if bn is not None else None, bn, self.activation == 'relu')
        x = self.linear(x)
        if self.normalization == 'batch':
            x = self.norm(x, epoch)
        elif self.normalization is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.act(x)
        return x


class MyConv2d(_FusedPointwise):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, activation=None,
                 momentum=0.1, normalization=None, bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self.activation = activation
        self.normalization = normalization
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        if normalization == 'batch':
            self.norm = MyBatchNorm2d(out_channels, momentum=momentum, affine=True,
                                      momentum_decay_step=bn_momentum_decay_step, momentum_decay=bn_momentum_decay)
        elif normalization == 'instance':
            self.norm = nn.InstanceNorm2d(out_channels, momentum=momentum, affine=True)
        act = _make_act(activation)
        if act is not None:
            self.act = act
        self.weight_init()

    def weight_init(self):
        ks = self.conv.kernel_size
        self.conv.weight.data.normal_(0, math.sqrt(2. / (ks[0] * ks[1] * self.conv.in_channels)))
        if self.conv.bias is not None:
            self.conv.bias.data.fill_(0)
        if self.normalization in ('batch', 'instance'):
            self.norm.weight.data.fill_(1)
            self.norm.bias.data.zero_()

    def forward(self, x, epoch=None):
        if self._fusable() and x.is_cuda:
            B, C, H, W = x.shape
            y, act_done = self._run(self._prep(x).view(B, C, H * W), None, epoch)
            y = y.view(B, -1, H, W)
        else:                                               # 3x3 decoder convs (UpConv): outside the hot path
            if not x.is_cuda and self._fusable():
                raise _ops.SonetHipError("MyConv2d(1x1) runs on the MI355X only (got a CPU tensor)")
            y, act_done = self.conv(x), False
            if self.normalization == 'batch':
                y = self.norm(y, epoch)
        if self.normalization is not None and self.normalization != 'batch':
            y = self.norm(y)
        if self.activation is not None and not act_done:
            y = self.act(y)
        return y


class UpConv(nn.Module):
    """Upsample x2 + 3x3 conv (models/layers.py:214-240); decoder only, stays on aten."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, output_padding=0, bias=True,
                 activation=None, normalization=None):
        super().__init__()
        self.activation = activation
        self.normalization = normalization
        self.up_sample = nn.Upsample(scale_factor=2)
        self.conv = MyConv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True,
                             activation=activation, normalization=normalization)
        self.weight_init()

    def weight_init(self):
        for m in self.modules():
            if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.fill_(0.001)

    def forward(self, x):
        return self.conv(self.up_sample(x))


class KNNModule(nn.Module):
    """Node-level grouping (models/layers.py:299-367): gather the K' neighbours of every SOM node
    (coordinates + features), de-centre, two fused 1x1 conv layers, max over the neighbourhood."""

    def __init__(self, in_channels, out_channels_list, activation, normalization, momentum=0.1,
                 bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self.layers = nn.ModuleList()
        prev = in_channels
        for c_out in out_channels_list:
            self.layers.append(MyConv2d(prev, c_out, kernel_size=1, stride=1, padding=0, bias=True,
                                        activation=activation, normalization=normalization, momentum=momentum,
                                        bn_momentum_decay_step=bn_momentum_decay_step,
                                        bn_momentum_decay=bn_momentum_decay))
            prev = c_out

    def forward(self, coordinate, x, precomputed_knn_I, K, center_type, epoch=None):
        coord = coordinate.detach()                                        # B x 3 x M
        if precomputed_knn_I is not None:
            assert precomputed_knn_I.size()[2] >= K
            knn_I = precomputed_knn_I[:, :, 0:K].contiguous()
        else:                                                              # fallback M x M kNN (layers.py:333-337)
            knn_I = _ops.knn_self(coord.float().contiguous(), K)
        if center_type not in ('avg', 'center'):
            raise ValueError(center_type)
        if (len(self.layers) == 2 and x.shape[1] % 16 == 0 and all(l._direct_ok(x) for l in self.layers)
                and self.layers[0].conv.in_channels == 3 + x.shape[1]):
            # no-grad h3 path: only the 3 de-centred coordinate rows are materialised (k-major); the features are gathered by
            # the first layer's operand loads; the max over the K neighbours reads k-major planes (coalesced)
            center, dec, gidx = _ops.knn_prepare(coord.float().contiguous(), knn_I, center_type == 'avg')
            l1, l2 = self.layers
            s1, t1 = l1._eval_affine()
            sdt = torch.bfloat16 if _ops.POINTMLP_PRECISION == "bf16" else torch.float32
            if _ops.NODE_LINEAR_SPLIT and _ops.x3_supported(x.shape[1], 0, l1.conv.out_channels):
                # W . cat(dec, x[:, gidx]) = (W_x . x)[:, gidx] + W_dec . dec: the 384-channel block once per node (M columns, not
                # K * M), the three coordinate channels as exact f32 fmas in the gather kernel
                wpf, wl = l1._packed_split(3)
                C1o = l1.conv.out_channels
                # (bf16 mode: z and h in bf16 storage -- z is rounded once more than in the gathered-layer form, inside the bf16 bound)
                z = _ops.pointmlp(x.to(sdt).contiguous(), wpf, _ops.const_vec(C1o, 1.0, x.device), _ops.const_vec(C1o, 0.0, x.device), False, C1o)
                h = _ops.node_gather_lead_affine_act(z, gidx, dec.float(), wl, s1, t1, l1.activation == 'relu')
            else:
                h = _ops.pointmlp(x.to(sdt).contiguous(), l1._packed_rotated(3), s1, t1, l1.activation == 'relu', l1.conv.out_channels,
                                  x2=dec.to(sdt), gidx=gidx)              # B x C1 x (K*M), k-major columns
            s2, t2 = l2._eval_affine()
            wp2 = l2._packed(h.shape[1], 0)
            # (the max over the K planes from the layer's epilogue -- sonet_hip.variants.pointmlp_kmax -- measured slower: variants build only)
            h = _ops.pointmlp(h, wp2, s2, t2, l2.activation == 'relu', l2.conv.out_channels)
            feature = _ops.planes_max(h, K)                               # B x C2 x M (storage type of h)
            return center, feature
        if not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32:
            # no-grad path: gathers, centre, de-centring and the concat in one kernel; max over K in one kernel
            center, h = _ops.knn_group(coord.contiguous(), x.contiguous(), knn_I, center_type == 'avg')
            for layer in self.layers:
                h = layer(h, epoch)
            return center, _ops.lastdim_max(h.contiguous())
        neighbors = operations.knn_gather_wrapper(coord, knn_I)            # B x 3 x M x K
        if center_type == 'avg':
            center = neighbors.mean(dim=3, keepdim=True)
        elif center_type == 'center':
            center = coord.unsqueeze(3)
        else:
            raise ValueError(center_type)
        decentered = (neighbors - center).detach()
        x_neighbors = operations.knn_gather_by_indexing(x, knn_I)          # B x C x M x K
        h = torch.cat((decentered, x_neighbors), dim=1)
        for layer in self.layers:
            h = layer(h, epoch)
        if torch.is_grad_enabled() and h.requires_grad:
            feature = _ops.lastdim_max_autograd(h)                     # autograd routes to ONE arg-max (the first), like the reference's torch.max
        else:
            feature = torch.amax(h, dim=3)                             # values only (torch.max also builds indices)
        return center.squeeze(3).detach(), feature


class PointNet(nn.Module):
    """Stack of EquivariantLayers, last one without norm / activation (models/layers.py:370-387)."""

    def __init__(self, in_channels, out_channels_list, activation, normalization, momentum=0.1,
                 bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self.layers = nn.ModuleList()
        prev = in_channels
        last = len(out_channels_list) - 1
        for i, c_out in enumerate(out_channels_list):
            if i != last:
                self.layers.append(EquivariantLayer(prev, c_out, activation, normalization, momentum,
                                                    bn_momentum_decay_step, bn_momentum_decay))
            else:
                self.layers.append(EquivariantLayer(prev, c_out, None, None))
            prev = c_out

    def forward(self, x, epoch=None):
        for layer in self.layers:
            x = layer(x, epoch)
        return x

    def forward_cat(self, lead, x, epoch=None):
        """forward(cat(lead, x)) for a narrow ``lead`` (the 3 centre rows of models/networks.py:191-194).  No-grad h3: the
        concat is replaced by a rotated weight pack (x first, lead as the second panel); otherwise the plain path."""
        if x.shape[1] % 16 == 0 and all(l._direct_ok(x) for l in self.layers):
            first = self.layers[0]
            s, t = first._eval_affine()
            sdt = torch.bfloat16 if _ops.POINTMLP_PRECISION == "bf16" else torch.float32
            h = _ops.pointmlp(x.to(sdt).contiguous(), first._packed_rotated(lead.shape[1]), s, t, first.activation == 'relu',
                              first.conv.out_channels, x2=lead.to(sdt).contiguous())
            for layer in self.layers[1:]:
                h = layer(h, epoch)
            return h
        return self.forward(torch.cat((lead, x), dim=1), epoch)


class PointResNet(nn.Module):
    """in -> c0 -> c1 -> ... -> c[k-2]; the last layer sees cat(c0 output, c[k-2] output)
    (models/layers.py:390-432).  The concat is never materialised: both panels go to the kernel."""

    def __init__(self, in_channels, out_channels_list, activation, normalization, momentum=0.1,
                 bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self.out_channels_list = out_channels_list
        self.layers = nn.ModuleList()
        prev = in_channels
        last = len(out_channels_list) - 1
        for i, c_out in enumerate(out_channels_list):
            if i != last:
                self.layers.append(EquivariantLayer(prev, c_out, activation, normalization, momentum,
                                                    bn_momentum_decay_step, bn_momentum_decay))
            else:
                self.layers.append(EquivariantLayer(prev + out_channels_list[0], c_out, None, None))
            prev = c_out

    def _fusable_eval(self, x):
        """One-kernel path: standard first-PointNet shape, eval BatchNorm + ReLU, no autograd, h3 arithmetic."""
        if self.training or torch.is_grad_enabled():             # training BN / autograd: layer-by-layer path
            return False
        # (the fused kernel computes in the fp16 split: "x3" must stay what the range guard falls back to -- f32 operand range
        #  end to end -- so it takes the four layer-wise launches)
        if _ops.POINTMLP_PRECISION not in ("h3", "bf16") or not _ops.FUSE_POINTRESNET or not x.is_cuda:
            return False
        if list(self.out_channels_list) != [64, 128, 256, 384] or x.shape[1] > 16:
            return False
        ls = self.layers
        if _ops.POINTMLP_PRECISION == "h3" and not all(l._h3_ok() for l in ls):     # weight side of the range guard: layer-wise (x3 where needed)
            return False
        hidden_ok = all(l.normalization == 'batch' and l.activation == 'relu' for l in ls[:3])
        return hidden_ok and ls[3].normalization is None and ls[3].activation is None and x.shape[2] * 384 * 4 < 4e9

    def _fused_state(self):
        ws = [l.conv.weight for l in self.layers]
        bf16 = _ops.POINTMLP_PRECISION == "bf16"
        key = tuple((w._version, w.data_ptr()) for w in ws) + (ws[0].device, bf16)
        if getattr(self, '_fused_key', None) != key:
            with torch.no_grad():
                pack = _ops.pointresnet_bf16_pack if bf16 else _ops.pointresnet_pack
                self._fused_w = pack(*[l._weight2d().detach().contiguous().float() for l in self.layers])
            self._fused_key = key
        aff = [l._eval_affine() for l in self.layers]                   # cached per layer
        akey = tuple(l._affine_gen for l in self.layers)                 # generation of each layer's folded (scale, shift)
        if getattr(self, '_fused_akey', None) != akey:
            self._fused_aff = torch.stack((torch.cat([a[0] for a in aff]), torch.cat([a[1] for a in aff])), dim=1).contiguous()
            self._fused_akey = akey
        return self._fused_w, self._fused_aff

    def pooled_sorted_ok(self, x, M):
        """Training, f32-class arithmetic, nobody reads first_pn_out: the hidden layers may run on NODE-SORTED columns and the last layer
        pool its own output (``sonet_pointmlp_h3_segpool_f32``).  Decided before any layer runs."""
        n = len(self.out_channels_list)
        last = self.layers[n - 1]
        if not (_ops.POINTMLP_PRECISION == "h3" and _ops.POOLED_TRAIN_EPILOGUE and _ops.H3_SEGPOOL and x.is_cuda and x.dtype == torch.float32):
            return False
        if n < 3 or last.normalization is not None or last.activation is not None or not last._fusable():
            return False
        c0, ck = self.out_channels_list[0], self.out_channels_list[n - 2]
        if c0 % 16 != 0 or last.conv.out_channels % 32 != 0 or not 0 < M <= 1024:
            return False
        # the pack the layer would pick: h3 (the weight side of the range guard may send it to x3: store + index_max then)
        return _ops.x3_supported(c0, ck, last.conv.out_channels) and last._h3_ok()

    def norm_on_load_ok(self, x):
        """f32-class training on node-sorted columns (``pooled_sorted_ok``): the hidden layers may hand their RAW outputs on and leave the
        normalise + ReLU pass to their consumers' operand loads.  Decided before any layer runs."""
        n = len(self.out_channels_list)
        if not (_ops.H3_NORM_ON_LOAD and _ops.WGRAD_KERNEL and _ops.STATS_EPILOGUE and _ops.POINTMLP_PRECISION == "h3" and x.dtype == torch.float32):
            return False
        for l in range(n - 1):
            lay = self.layers[l]
            if not (lay.normalization == 'batch' and lay.norm.training and lay.activation == 'relu' and lay._fusable()):
                return False
            if l >= 1 and not (lay._h3_ok() and _ops.xaff_ok(lay.conv.in_channels, 0, lay.conv.out_channels)
                               and x.shape[0] * x.shape[2] * lay.conv.out_channels * 4 >= (32 << 20)):
                return False
        last = self.layers[n - 1]
        return _ops.xaff_ok(self.out_channels_list[0], self.out_channels_list[n - 2], last.conv.out_channels)

    def bf16_norm_on_load_ok(self, x, M):
        """bf16 training, nobody reads first_pn_out (the pool is the last layer's epilogue): the hidden layers hand their RAW bf16 outputs on
        and every consumer -- next layer, pooled last layer, weight gradients -- normalises in its operand path.  Decided before any layer
        runs; the original column order (no sorted copy needed: the bf16 pool epilogue keeps its bins in LDS)."""
        n = len(self.out_channels_list)
        if not (_ops.BF16_NORM_ON_LOAD and _ops.WGRAD_KERNEL and _ops.STATS_EPILOGUE and _ops.POOLED_TRAIN_EPILOGUE
                and _ops.POINTMLP_PRECISION == "bf16" and x.is_cuda):
            return False
        B, _, L = x.shape
        for l in range(n - 1):
            lay = self.layers[l]
            if not (lay.normalization == 'batch' and lay.norm.training and lay.activation == 'relu' and lay._fusable()):
                return False
            if l >= 1 and not _ops.bf16_xaff_ok(B, lay.conv.in_channels, 0, lay.conv.out_channels, L):
                return False
        C1, C2, Cout = self.out_channels_list[0], self.out_channels_list[n - 2], self.layers[n - 1].conv.out_channels
        if L % 2 or L > 65534 or (C1 + C2) % 64 or C1 % 16 or Cout % 32 or not 0 < M <= 255 or B > 65535:
            return False
        return _ops.wgrad_bf16_xaff_ok(B, Cout, C1, L)

    def forward_pooled(self, x, min_idx_i32, row_max, M, epoch=None, need_dense=True, pos0=None):
        """Training path of the encoder: hidden layers as usual, then the last layer and the per-node arg-max pool as one
        autograd node -> (first_pn_out, first_pn_out_masked_max, gather_index) or None when the layout does not allow it.
        need_dense=False (nobody reads first_pn_out: classifier, autoencoder): where the arithmetic has the kernel for it (bf16) the
        first element is None -- the tensor is never written."""
        n = len(self.out_channels_list)
        last = self.layers[n - 1]
        if n < 3 or last.normalization is not None or last.activation is not None or not last._fusable():
            return None
        if self.out_channels_list[0] % 16 != 0:                          # decided BEFORE any layer runs (training BN must not run twice)
            return None
        if pos0 is not None and self.norm_on_load_ok(x):
            # the normalised activations of the hidden layers are never written: every consumer (next layer, weight gradients, the pooled
            # last layer and its sparse weight gradient) normalises on load
            h, _ = self.layers[0]._run(_FusedPointwise._prep(x), None, epoch, defer=True)
            skip = h
            # the first layer's output feeds the second layer and the last one: the last layer's input gradient (computed first) is added
            # by the store of the second layer's input-gradient launch, not by a pass of autograd's (``_GradCarry``)
            carry = _GradCarry() if (_ops.GRAD_CARRY and _ops._graph_task_id is not None and torch.is_grad_enabled() and skip[0].requires_grad) else None
            for l in range(1, n - 1):
                h, _ = self.layers[l]._run(h[0], None, epoch, xaff=(h[1], h[2], True), defer=True, carry=carry if l == 1 else None)
            wp = last._packed(skip[0].shape[1], h[0].shape[1])
            if wp.dtype == torch.int8:
                return _PooledLastLayerFn.apply(skip[0], h[0], last._weight2d(), last._bias(), wp, min_idx_i32, row_max, M, False, pos0,
                                                (skip[1], skip[2], True, h[1], h[2], True), carry)
            skip, t = _Materialise.apply(skip[0], skip[1], skip[2], True), _Materialise.apply(h[0], h[1], h[2], True)
            return _PooledLastLayerFn.apply(skip.contiguous(), t.contiguous(), last._weight2d(), last._bias(), wp, min_idx_i32, row_max, M,
                                            bool(need_dense), pos0, None, carry)
        elif pos0 is None and not need_dense and self.bf16_norm_on_load_ok(x, M):
            # bf16: the same data flow in the original column order (round 6) -- the three normalise + ReLU passes over B x C x kN are gone
            h, _ = self.layers[0]._run(_FusedPointwise._prep(x), None, epoch, defer=True)
            skip = h
            carry = _GradCarry() if (_ops.GRAD_CARRY and _ops._graph_task_id is not None and torch.is_grad_enabled() and skip[0].requires_grad) else None
            for l in range(1, n - 1):
                h, _ = self.layers[l]._run(h[0], None, epoch, xaff=(h[1], h[2], True), defer=True, carry=carry if l == 1 else None)
            wp = last._packed(skip[0].shape[1], h[0].shape[1])
            if wp.dtype == torch.int16 and _ops.pointmlp_bf16_pool_ok(skip[0], h[0], last.conv.out_channels, M):
                return _PooledLastLayerFn.apply(skip[0], h[0], last._weight2d(), last._bias(), wp, min_idx_i32, row_max, M, False, None,
                                                (skip[1], skip[2], True, h[1], h[2], True), carry)
            skip, t = _Materialise.apply(skip[0], skip[1], skip[2], True), _Materialise.apply(h[0], h[1], h[2], True)
            return _PooledLastLayerFn.apply(skip.contiguous(), t.contiguous(), last._weight2d(), last._bias(), wp, min_idx_i32, row_max, M,
                                            bool(need_dense), None, None, carry)
        else:
            skip = self.layers[0](x, epoch)
            t = skip
            for l in range(1, n - 1):
                t = self.layers[l](t, epoch)
            wp = last._packed(skip.shape[1], t.shape[1])
        return _PooledLastLayerFn.apply(skip.contiguous(), t.contiguous(), last._weight2d(), last._bias(), wp, min_idx_i32, row_max, M, bool(need_dense), pos0)

    def forward(self, x, epoch=None):
        self.last_p16 = None
        if self._fusable_eval(x):
            wstream, affine = self._fused_state()
            if _ops.POINTMLP_PRECISION == "bf16":
                return _ops.pointresnet_bf16(x.float().contiguous(), wstream, affine)
            if getattr(self, "emit_p16", False) and _ops.P16_CHAINS:
                # a caller that feeds the output to a third-generation layer (the segmenter) takes it pre-split from the same launch
                only = self.emit_p16 == "only" and not torch.is_grad_enabled()
                y, self.last_p16 = _ops.pointresnet_fused(_FusedPointwise._prep(x).float(), wstream, affine, want_p16="only" if only else True)
                return y                                   # (None when only the planes were written)
            return _ops.pointresnet_fused(_FusedPointwise._prep(x).float(), wstream, affine)
        n = len(self.out_channels_list)
        skip = self.layers[0](x, epoch)
        t = skip
        for l in range(1, n - 1):
            t = self.layers[l](t, epoch)
        if skip.shape[1] % 8 == 0:
            return self.layers[n - 1](t, epoch, x_skip=skip)              # fused concat (skip first)
        return self.layers[n - 1](torch.cat((skip, t), dim=1), epoch)
