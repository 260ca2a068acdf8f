"""One-process-per-GPU data parallelism for the SO-Net hot path.

The path shards over the batch: every cloud (its points, its M nodes, its kNN table) is independent
in the forward pass, so inference needs NO data-path collective -- each rank runs its own shard and
throughput adds up ("weak" scaling).  Training adds exactly one exchange per step: a gradient
all-reduce between ``loss.backward()`` and the optimizer steps of the task ``Model`` shells
(models/classifier.py:95-99).  The reference has no distributed code at all; this is new.

Design for MI355X / xGMI: the whole gradient payload (Encoder 1,999,041 + Classifier 667,944
parameters = 10.7 MB fp32) lives in ONE flat buffer cut into at most three ~4 MB buckets, each one
RCCL all-reduce started from a gradient hook as soon as backward has filled it.  xGMI is
point-to-point (7 links x ~153 GB/s per GPU), so messages of this size are latency/algorithm-bound:
per-parameter calls would only multiply that latency, a few large buckets keep it while the
transfer of the heads' gradients hides behind the first PointNet's backward.  Parameters whose ``.grad``
is None (the never-called ``transformer.*``, models/networks.py:78) are skipped, which is why plain
DistributedDataParallel (which expects every registered parameter to take part) is not used.
BatchNorm statistics stay per rank, as in the reference (no SyncBN).

Backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` on CPU (tests).  Rendezvous via the
MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE / LOCAL_RANK environment (torch.distributed.run).
"""
import collections
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None, force=False):
    """Join the process group described by the environment.  Returns (world, rank, local_rank).
    ``force``: create the group even for a single rank (so that a 1-GPU box still exercises RCCL)."""
    world, rank, local_rank = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_range(global_batch, world, rank):
    """Contiguous [lo, hi) slice of the global batch owned by ``rank`` (remainder to the low ranks)."""
    base, rem = divmod(int(global_batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors, world, rank):
    """Slice every B-leading tensor of a dict / tuple to this rank's shard."""
    def cut(t):
        lo, hi = shard_range(t.shape[0], world, rank)
        return t[lo:hi].contiguous()
    if isinstance(tensors, dict):
        return {k: cut(v) for k, v in tensors.items()}
    return type(tensors)(cut(v) for v in tensors)


@torch.no_grad()
def broadcast_parameters(modules, src=0):
    """Make every rank start from rank ``src``'s parameters and buffers (the reference's init is
    unseeded random).  One flat broadcast per dtype."""
    if world_size() == 1:
        return
    by_dtype = {}
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class GradientAllReducer:
    """Bucketed gradient averaging for a set of modules (two optimizers, dead parameters OK), overlapped with backward.

    ``reduce()`` is called between ``loss.backward()`` and ``optimizer.step()``.  The set of parameters that take part
    is decided on the first call (those with ``grad is not None``) and checked to be identical on every rank; that
    first step is one flat all-reduce.  From the second step on, the live parameters sit in ``len(buckets)`` slices of
    one flat buffer, cut in REVERSE registration order (the order backward produces gradients in: heads first, the
    first PointNet last) at ``bucket_bytes``; a post-accumulate hook copies each gradient into its slice and, when a
    bucket is complete, starts its all-reduce asynchronously -- the heads' and node-level layers' gradients (about 9 of
    the 10.7 MB) travel over xGMI while the first PointNet's backward, the bulk of the step, is still running.
    ``reduce()`` then only waits, scales and scatters back.  Few large buckets on purpose: xGMI is point-to-point and a
    message of a few MB is latency-bound (module docstring).  ``overlap=False`` keeps the single flat call.
    """

    def __init__(self, modules, bucket_bytes=4 << 20, overlap=True, always_reduce=False):
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        self.bucket_bytes = int(bucket_bytes)
        self.overlap = bool(overlap)
        self.always_reduce = bool(always_reduce)      # run the collectives even in a single-rank group (tests on one GPU)
        self._next_launch = 0        # buckets are launched strictly in index order (same order on every rank)
        self._dirty = False          # a second backward touched a bucket that was already on the wire
        self._sync = True            # False inside no_sync(): gradient accumulation, hooks stay quiet
        self._exposed_events = collections.deque(maxlen=256)   # (start, end) HIP events around the waits of the last reduce() calls
        self._live = None            # indices into self.params, flat-buffer order
        self._flat = None
        self._slices = {}            # param index -> (offset, numel)
        self.buckets = []            # [(offset, numel, [param indices])]
        self._bucket_of = {}
        self._pending = []           # per bucket: gradients still missing this step
        self._work = []              # per bucket: async handle (or None)
        self._work_done = []
        self._hooks = []
        self._armed = False

    def _setup(self):
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if world_size() > 1:
            sig = torch.tensor([len(live), sum(live) % (2 ** 31)], dtype=torch.int64, device=self.params[0].device)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not (torch.equal(lo, sig) and torch.equal(hi, sig)):
                raise RuntimeError("ranks disagree on which parameters received gradients")
        live = live[::-1]                                            # backward order: last registered first
        self._live = live
        ref = self.params[live[0]] if live else self.params[0]
        off, start, members = 0, 0, []
        for i in live:
            n = self.params[i].numel()
            self._slices[i] = (off, n)
            members.append(i)
            off += n
            if (off - start) * ref.element_size() >= self.bucket_bytes:
                self.buckets.append((start, off - start, members))
                start, members = off, []
        if members:
            self.buckets.append((start, off - start, members))
        for b, (_, _, members) in enumerate(self.buckets):
            for i in members:
                self._bucket_of[i] = b
        self._flat = torch.zeros(off, dtype=ref.dtype, device=ref.device)
        # views of the flat buffer shaped like the parameters, per bucket: gradients go in and out with ONE multi-tensor copy per
        # bucket (a copy per parameter was 110 four-microsecond launches per training step)
        self._views = [[self._flat[self._slices[i][0]:self._slices[i][0] + self._slices[i][1]].view_as(self.params[i]) for i in members]
                       for _, _, members in self.buckets]
        if self.overlap and (world_size() > 1 or self.always_reduce):
            for i in live:
                self._hooks.append(self.params[i].register_post_accumulate_grad_hook(self._make_hook(i)))
            self._arm()

    def _arm(self):
        self._pending = [len(members) for _, _, members in self.buckets]
        self._work = [None] * len(self.buckets)
        self._next_launch = 0
        self._dirty = False
        self._armed = True

    def _make_hook(self, i):
        def hook(p):
            if not self._armed or not self._sync or p.grad is None:
                return
            b = self._bucket_of[i]
            if self._pending[b] <= 0:
                # a SECOND backward before reduce() (two losses, accumulation without no_sync()): what is on the wire for
                # this bucket is a partial sum -- reduce() redoes the exchange from the accumulated .grad tensors
                self._dirty = True
                return
            self._pending[b] -= 1
            # launch every complete bucket up to the first incomplete one, in index order: a rank whose autograd graph
            # finishes bucket 2 before bucket 1 still issues the collectives in the order every other rank does
            while self._next_launch < len(self.buckets) and self._pending[self._next_launch] == 0:
                bo, bn, members = self.buckets[self._next_launch]
                _join_side_streams()                              # (weight gradients still in flight on a side stream: sonet_hip.ops.side_stream.join(defer=True))
                _copy_all(self._views[self._next_launch], [self.params[j].grad for j in members])
                self._work[self._next_launch] = dist.all_reduce(self._flat[bo:bo + bn], op=dist.ReduceOp.SUM, async_op=True)
                self._next_launch += 1
        # (models/layers.py ``_grad_slot_empty``: this hook joins the side streams before it reads a gradient, so a weight that carries it may
        #  still leave its weight-gradient launch un-joined until the hook -- or the end of the pass -- comes)
        hook._sonet_joins_side_streams = True
        return hook

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only accumulate into ``.grad``; the
        exchange happens on the first backward + ``reduce()`` outside (which sends the accumulated gradients)."""
        red = self

        class _NoSync:
            def __enter__(self_inner):
                red._sync = False

            def __exit__(self_inner, *exc):
                red._sync = True
                return False
        return _NoSync()

    def exposed_ms(self, last=None):
        """Mean GPU time per ``reduce()`` (over the last ``last`` calls) between "backward has queued its last kernel" and
        "every bucket has arrived": the part of the gradient exchange that was NOT hidden behind backward.  Synchronises."""
        evs = list(self._exposed_events)
        evs = evs[-last:] if last else evs
        if not evs:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks, self._armed = [], False

    @torch.no_grad()
    def reduce(self):
        first = self._live is None
        if first:
            self._setup()
        w = world_size()
        if (w == 1 and not self.always_reduce) or not self._live:
            return 0
        ev = None
        if self._flat.is_cuda:                                            # GPU time the compute stream spends waiting for the collectives
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        overlapped = self._armed and not first and not self._dirty
        if self._armed and not first and self._dirty:
            for wk in self._work:                                         # drain what the first backward put on the wire
                if wk is not None:
                    wk.wait()
        if overlapped:
            for b, (bo, bn, members) in enumerate(self.buckets):
                if self._pending[b] != 0:
                    missing = [i for i in members if self.params[i].grad is None]
                    raise RuntimeError("bucket %d: %d gradient(s) did not arrive during backward (parameters %s had one on "
                                       "the first step)" % (b, self._pending[b], missing[:4]))
                self._work[b].wait()
        else:
            for i in self._live:
                if self.params[i].grad is None:
                    raise RuntimeError("parameter %d had a gradient on the first step but has none now" % i)
            for b, (_, _, members) in enumerate(self.buckets):
                _copy_all(self._views[b], [self.params[j].grad for j in members])
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
        if ev is not None:
            ev[1].record()
            self._exposed_events.append(ev)
        self._flat.div_(w)
        for b, (_, _, members) in enumerate(self.buckets):
            _copy_all([self.params[j].grad for j in members], self._views[b])
        self._work_done = list(self._work)                            # (what the hooks launched during this step: tests look at it)
        if self._armed:
            self._arm()                                               # next backward
        return self._flat.numel() * self._flat.element_size()


def _join_side_streams():
    from . import ops
    ops.join_side_streams()


def _copy_all(dst, src):
    """dst[k].copy_(src[k]) for all k in one multi-tensor launch where torch has it."""
    if not dst:
        return
    try:
        torch._foreach_copy_(dst, src)
    except (AttributeError, RuntimeError):
        for d, s_ in zip(dst, src):
            d.copy_(s_)


def all_reduce_max(value, device):
    """MAX over ranks of a python float (bench timing contract)."""
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world_size() > 1:
        if dist.get_backend() == "nccl":
            # name the device: without it the first barrier guesses one from the rank ("devices used by this process
            # are currently unknown"), which is wrong whenever the visible-device list is remapped
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def shutdown():
    """Leave the process group (no-op for a single process); ranks call it right before they return."""
    if dist.is_initialized():
        dist.destroy_process_group()
