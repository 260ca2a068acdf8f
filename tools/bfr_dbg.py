import os, sys
sys.path.insert(0, "so-net_amd")
import torch
from sonet_hip import ops
DEV="cuda:0"
def run(B,C1,C2,Cout,L,relu):
    g = torch.Generator().manual_seed(C1 + 3 * Cout + L)
    x1 = torch.randn(B, C1, L, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, C2, L, generator=g).to(torch.bfloat16).to(DEV) if C2 else None
    W = (torch.randn(Cout, C1 + C2, generator=g) * (2.0 / (C1 + C2)) ** 0.5).to(DEV)
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    wp = ops.pointmlp_pack(W, "bf16")
    y = ops.pointmlp(x1, wp, scale, shift, relu, Cout, x2=x2)
    parts = [ops.pointmlp(x1[b0:b0 + 16].contiguous(), wp, scale, shift, relu, Cout, x2=x2[b0:b0 + 16].contiguous() if C2 else None) for b0 in range(0, B, 16)]
    ref = torch.cat(parts, dim=0)
    bad = (y != ref)
    n = int(bad.sum())
    print((B,C1,C2,Cout,L), "mismatches", n, "of", y.numel())
    if n:
        idx = bad.nonzero()
        print(" first", idx[:5].tolist(), "last", idx[-3:].tolist())
        print(" clouds", sorted(set(idx[:,0].tolist()))[:10], " channels", sorted(set(idx[:,1].tolist()))[:10], "n", len(set(idx[:,1].tolist())))
        cols = idx[:,2]
        print(" cols min/max", int(cols.min()), int(cols.max()), " col%64 set", sorted(set((cols%64).tolist()))[:12])
        i=idx[0]; print(" got", float(y[i[0],i[1],i[2]]), "ref", float(ref[i[0],i[1],i[2]]))
for sh in [(36,64,0,128,15000,True),(64,128,0,256,8192,True),(36,64,256,384,15000,False),(36,128,0,64,15000,False)]:
    run(*sh)
