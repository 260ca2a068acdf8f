"""Worker of tests/test_gpu_round2.py::test_gradient_all_reducer_on_rccl -- one rank per visible GPU under
torch.distributed.run, backend nccl (= RCCL).  A classifier training step on this rank's shard; the hook-driven bucketed
all-reduce must leave every rank with the average of the per-rank gradients (checked against an all_gather of them)."""
import os
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "so-net_amd"), ROOT):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from models import networks as NW
    from sonet_hip import dp, synth
    world, rank, local_rank = dp.init_distributed(backend="nccl", force=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    B, N = 4, 512
    opt = Namespace(gpu_id=local_rank, device=dev, batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.7, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), 100 + rank)                  # ranks start different on purpose ...
    synth.fill_state_dict_(cls.state_dict(), 200 + rank)
    enc.to(dev).train()
    cls.to(dev).train()
    dp.broadcast_parameters([enc, cls])                                    # ... and are made equal over RCCL
    flat = torch.cat([p.detach().reshape(-1) for p in list(enc.parameters()) + list(cls.parameters())])
    ref0 = flat.clone()
    dist.broadcast(ref0, src=0)
    assert torch.equal(flat, ref0), "broadcast_parameters left the ranks different"
    inp = synth.make_inputs(B, N, seed=300 + rank, device=dev)
    red = dp.GradientAllReducer([enc, cls], always_reduce=True)
    torch.manual_seed(1)                                                   # same dropout masks everywhere (not required)

    def backward():
        enc.zero_grad(set_to_none=True)
        cls.zero_grad(set_to_none=True)
        score = cls(enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=True, epoch=0), 0)
        torch.nn.functional.cross_entropy(score, inp["label"]).backward()

    for step in range(3):                                                  # step 0: flat call; 1, 2: hook-driven buckets
        backward()
        live = [p for p in list(enc.parameters()) + list(cls.parameters()) if p.grad is not None]
        local = torch.cat([p.grad.reshape(-1) for p in live]).clone()
        nbytes = red.reduce()
        got = torch.cat([p.grad.reshape(-1) for p in live])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = torch.stack(gathered).sum(0) / world
        assert nbytes == local.numel() * 4, (nbytes, local.numel())
        err = (got - want).abs().max().item()
        assert err <= 1e-6 * max(1.0, want.abs().max().item()), (step, err)
        if step >= 1:
            assert all(w is not None for w in red._work_done), "a bucket was not started from a gradient hook"
    # The reducer's hooks sit on every weight: they join the side streams before they read a gradient, so the weight-gradient launches of the
    # layers may still stay un-joined until then (models/layers.py _grad_slot_empty) -- the step must not change by a bit against per-layer joins.
    from models import layers as L
    from sonet_hip import ops
    real, verdicts, res = L._grad_slot_empty, [], {}
    L._grad_slot_empty = lambda w, t=None: (verdicts.append(real(w, t)) or verdicts[-1])
    old = (ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL)
    ops.H3_SEGPOOL = False                                                 # (the bit-reproducible form of the pooled layer)
    try:
        for flag in (True, False):
            ops.DEFER_WGRAD_JOIN = flag
            torch.manual_seed(7)                                           # the same dropout masks in both passes
            del verdicts[:]
            backward()
            red.reduce()
            res[flag] = (torch.cat([p.grad.reshape(-1) for p in live]).clone(), sum(verdicts))
    finally:
        L._grad_slot_empty = real
        ops.DEFER_WGRAD_JOIN, ops.H3_SEGPOOL = old
    assert res[True][1] >= 4, "no weight deferred its join although only the reducer's own hooks are attached (%d)" % res[True][1]
    assert torch.equal(res[True][0], res[False][0]), "deferred joins under the reducer's hooks changed the gradients"
    dead = [n for n, p in enc.named_parameters() if p.grad is None]
    assert dead and all(n.startswith("transformer.") for n in dead), dead[:3]
    ms = red.exposed_ms()
    dp.barrier()
    if rank == 0:
        print("RCCL_OK world=%d buckets=%d bytes=%d exposed_ms=%.3f" % (world, len(red.buckets), nbytes, ms))
    dp.shutdown()


if __name__ == "__main__":
    main()
