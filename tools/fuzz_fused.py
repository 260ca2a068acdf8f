"""tools/fuzz_fused.py [n] [seed] -- random (B, N, M, k, node layout) for the fused first PointNet: the pooled kernel (node-sorted
input, per-node max-pool in the epilogue) against the store kernel + index_max_gather on the same weights (1e-6 of max(|ref|, rms)),
the two-launch SOM stage against the separate launches (bit-exact), and the store kernel against the four layer launches."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "so-net_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from models import layers as Lm  # noqa: E402
from sonet_hip import ops, synth  # noqa: E402

DEV = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
synth.fill_state_dict_(pr.state_dict(), seed=7)
pr.to(DEV).eval()
ops.POINTMLP_PRECISION, ops.FUSE_POINTRESNET = "h3", True
wstream, affine = pr._fused_state()
bad = 0
for it in range(n):
    B = rng.choice([1, 2, 3, 5])
    N = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 85, 86, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 777, 1024, 1500, 2049, 5000])
    M = rng.choice([1, 2, 4, 9, 16, 36, 64, 100])
    k = rng.choice([1, 2, 3]) if M >= 3 else 1
    kind = rng.choice(["uniform", "som"]) if (M == 64 and N >= 64) else "uniform"
    inp = synth.make_inputs(B, N, M=M, som_k=min(9, M), seed=1000 + it, node_kind=kind)
    x, sn, node = inp["pc"].to(DEV), inp["sn"].to(DEV), inp["node"].to(DEV)
    why = []
    a = ops.som_assign(x, node, k)
    g = ops.som_group(x, sn, a, want_augmented=True)
    s = ops.som_sort_group(x, sn, a)
    a2, s2 = ops.som_assign_sort(x, sn, node, k)
    for key in ("ids_sorted", "node_off", "som_node", "count"):
        if key in s and key in s2 and not torch.equal(s[key], s2[key]):
            why.append("som_assign_sort %s differs" % key)
    # inside a node's run the order of the point copies is the kernel's own: compare the runs as multisets of columns, and the
    # column of original copy 0 by value
    xa, xb, ids = s["x_aug_sorted"].cpu(), s2["x_aug_sorted"].cpu(), s["ids_sorted"].cpu().long()
    for b_ in range(B):
        key_a = torch.cat([ids[b_].double().view(1, -1), xa[b_].double()], 0).t()
        key_b = torch.cat([ids[b_].double().view(1, -1), xb[b_].double()], 0).t()
        sa = key_a[np.lexsort(key_a.numpy().T[::-1])]
        sb = key_b[np.lexsort(key_b.numpy().T[::-1])]
        if not torch.equal(sa, sb):
            why.append("som_assign_sort: cloud %d holds other columns" % b_)
        if not torch.equal(xa[b_, :, int(s["pos0"][b_])], xb[b_, :, int(s2["pos0"][b_])]):
            why.append("som_assign_sort: pos0 of cloud %d points at another column" % b_)
    if not torch.equal(a.min_idx_i32, a2.min_idx_i32):
        why.append("som_assign_sort ids differ")
    with torch.no_grad():
        first = pr(g["x_augmented"])
        _, ref = ops.index_max_gather(first, a.min_idx_i32, M, g["row_max"])
        got = ops.pointresnet_fused_pool(s2, wstream, affine, M)
        old = ops.FUSE_POINTRESNET
        ops.FUSE_POINTRESNET = False
        layerwise = pr(g["x_augmented"])
        ops.FUSE_POINTRESNET = old
    r = ref.double()
    tol = 1e-6 * max(float(r.abs().max()), float(r.pow(2).mean().sqrt()), 1e-30)
    err = float((got.double() - r).abs().max())
    if not err <= tol:
        why.append("pool vs store + index_max: %.3g > %.3g" % (err, tol))
    lw = layerwise.double()
    err2 = float((first.double() - lw).abs().max())
    tol2 = 2e-6 * max(float(lw.abs().max()), 1e-30)
    if not err2 <= tol2:
        why.append("store vs layerwise: %.3g > %.3g" % (err2, tol2))
    if why:
        bad += 1
        print("case %d B=%d N=%d M=%d k=%d %s: %s" % (it, B, N, M, k, kind, "; ".join(why)), flush=True)
print("%d cases, %d bad" % (n, bad))
sys.exit(1 if bad else 0)
