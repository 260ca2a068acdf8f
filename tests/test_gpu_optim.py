"""sonet_hip.optim.FusedAdam (one launch per step) against torch.optim.Adam on the same parameters and gradients."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 6, 1), (64,), (128, 64, 1), (384, 320, 1), (1024, 768, 1, 1), (5,), (1,), (40, 256), (4097,)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]


def test_fused_adam_matches_torch_adam_over_several_steps():
    from sonet_hip.optim import FusedAdam
    pa, pb = _params(0, DEV), _params(0, DEV)
    oa = torch.optim.Adam(pa, lr=1e-3, betas=(0.9, 0.999))
    ob = FusedAdam(pb, lr=1e-3, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(1)
    for step in range(7):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if (i == 5 and step % 2 == 0) or (i == 2 and step == 3):        # a parameter without a gradient: skipped, its step count stays
                a.grad = b.grad = None
                continue
            gr = (torch.randn(a.shape, generator=g) * (10.0 ** ((i % 3) - 2))).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            err = float((a.detach() - b.detach()).abs().max())
            assert err <= 2e-6 * max(1.0, float(a.detach().abs().max())), (step, i, err)
    for a, b in zip(pa, pb):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"])
        # (torch's kernels contract a + b * c into an fma, this one rounds the product first: a few ulp over seven steps)
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 5e-6 * max(1e-6, float(sa["exp_avg"].abs().max()))
        assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 5e-6 * max(1e-12, float(sa["exp_avg_sq"].abs().max()))


def test_fused_adam_state_dict_round_trip_and_errors():
    from sonet_hip.optim import FusedAdam
    from sonet_hip.ops import SonetHipError
    pa = _params(3, DEV)
    o1 = FusedAdam(pa, lr=2e-3)
    for p in pa:
        p.grad = torch.ones_like(p)
    o1.step()
    import copy
    sd = copy.deepcopy(o1.state_dict())          # (load_state_dict does not copy tensors that already match the parameter's dtype / device)
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    o2 = FusedAdam(pb, lr=2e-3)
    o2.load_state_dict(sd)
    for p, q in zip(pa, pb):
        p.grad = torch.full_like(p, 0.5)
        q.grad = torch.full_like(q, 0.5)
    o1.step()
    o2.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p.detach(), q.detach())
    with pytest.raises(SonetHipError):
        o = FusedAdam([torch.nn.Parameter(torch.zeros(3))])
        o.param_groups[0]["params"][0].grad = torch.zeros(3)
        o.step()
    with pytest.raises(ValueError):
        FusedAdam(pa, lr=-1.0)
    v0 = pa[0]._version
    pa[0].grad = torch.ones_like(pa[0])
    o1.step()
    assert pa[0]._version > v0                   # caches keyed on the version (packed weights) see the update


def test_fused_adam_load_state_dict_after_a_step_takes_the_loaded_moments():
    """the launch plan (pointers into the flat moment buffers) is built at the first step; a load_state_dict() AFTER a step
    must drop it, or the kernel keeps updating buffers nobody reads and state_dict() returns stale moments."""
    import copy
    from sonet_hip.optim import FusedAdam
    from sonet_hip.ops import SonetHipError
    pa = _params(11, DEV)
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = torch.optim.Adam(pa, lr=1e-3), FusedAdam(pb, lr=1e-3)
    g = torch.Generator().manual_seed(2)

    def grads():
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
    for _ in range(3):
        grads()
        oa.step()
        ob.step()
    snap_a, snap_b = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    snap_p = [p.detach().clone() for p in pb]
    for _ in range(2):                                        # move on, then go back to the snapshot
        grads()
        oa.step()
        ob.step()
    oa.load_state_dict(snap_a)
    ob.load_state_dict(snap_b)
    with torch.no_grad():
        for a, b, s in zip(pa, pb, snap_p):
            a.copy_(s)
            b.copy_(s)
    grads()
    oa.step()
    ob.step()
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * max(1.0, float(a.detach().abs().max()))
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 4.0
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 5e-6 * max(1e-6, float(sa["exp_avg"].abs().max()))
    # the moments state_dict() hands out are the ones the kernel updates
    grads()
    before = ob.state_dict()["state"][0]["exp_avg"].clone()
    ob.step()
    assert not torch.equal(before, ob.state_dict()["state"][0]["exp_avg"])
    # a state edited by hand behind the plan is an error, not a silent no-op
    ob.state[pb[0]]["exp_avg"] = torch.zeros_like(pb[0])
    with pytest.raises(SonetHipError):
        ob.step()
    # add_param_group after a step: the new group is stepped too
    extra = torch.nn.Parameter(torch.ones(7, device=DEV))
    oc = FusedAdam([pb[1]], lr=1e-2)
    pb[1].grad = torch.ones_like(pb[1])
    oc.step()
    oc.add_param_group({"params": [extra]})
    extra.grad = torch.ones_like(extra)
    pb[1].grad = torch.ones_like(pb[1])
    oc.step()
    assert float(extra.detach().max()) < 1.0
