"""tools/bench_pool_epilogue.py -- the last layer of the first PointNet in bf16 training (320 -> 384 at 64 x 15000 columns): the storing launch +
index_max_gather on the stored tensor against the launch whose epilogue is the pool (sonet_pointmlp_bf16_pool).  Warm, back-to-back."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import ops  # noqa: E402

DEV = "cuda:0"


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


B, C1, C2, Cout, L, M = 64, 64, 256, 384, 15000, 64
g = torch.Generator().manual_seed(0)
x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV)
W = (torch.randn(Cout, C1 + C2, generator=g) * (C1 + C2) ** -0.5).to(DEV)
bias = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
ids = torch.randint(0, M, (B, L), generator=g, dtype=torch.int32).to(DEV)
ids_sorted = torch.sort(ids, dim=1).values.contiguous()
row_max = torch.ones(B, M, dtype=torch.int32, device=DEV)
wp = ops.pointmlp_pack(W, "bf16")
one = ops.const_vec(Cout, 1.0, DEV)
a = t(lambda: ops.pointmlp(x1, wp, one, bias, False, Cout, x2=x2))
y = ops.pointmlp(x1, wp, one, bias, False, Cout, x2=x2)
b = t(lambda: ops.index_max_gather(y, ids, M, row_max))
c = t(lambda: ops.pointmlp_bf16_pool(x1, wp, one, bias, False, Cout, ids, M, row_max, x2=x2))
d = t(lambda: ops.pointmlp_bf16_pool(x1, wp, one, bias, False, Cout, ids_sorted, M, row_max, x2=x2))
for ns in ("3", "4", "6"):
    os.environ["SONET_BF16_POOL_NS"] = ns
    print("  (variants library) %s output slabs: %.1f us" % (ns, t(lambda: ops.pointmlp_bf16_pool(x1, wp, one, bias, False, Cout, ids, M, row_max, x2=x2))))
os.environ.pop("SONET_BF16_POOL_NS", None)
for k, v, what in (("SONET_BF16_POOL_ABL", "1", "no epilogue at all"), ("SONET_BF16_POOL_ABL", "2", "epilogue arithmetic, no bin traffic beyond the shadow reads"),
                   ("SONET_BF16_SYNC", "0", "no workgroup barriers"), ("SONET_BF16_SYNC", "1", "one barrier per column group")):
    os.environ[k] = v
    print("  (variants library) %s: %.1f us" % (what, t(lambda: ops.pointmlp_bf16_pool(x1, wp, one, bias, False, Cout, ids, M, row_max, x2=x2))))
    os.environ.pop(k, None)
print("layer (store) %.1f us + index_max_gather %.1f us = %.1f us   |   layer with pool epilogue: random node order %.1f us, node-sorted columns %.1f us"
      % (a, b, a + b, c, d))
