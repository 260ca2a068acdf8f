"""tools/fused_variants.py NAME [NAME ...] -- time experimental builds of the fused first-PointNet kernel
(tools/build_variant.sh NAME flags -> so-net_amd/lib/variants/libsonet_hip_NAME.so) against each other on the bench
shape (B=64, 5000 points, 8x8 SOM) and check that every variant's output is bit-identical to the first one's.
One child process per variant (the library handle is process-wide); each child runs under a hard timeout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(name, out_path):
    sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
    import torch
    from sonet_hip import _lib
    libname, _, abl = name.partition("@")                      # NAME@64: the variant with SONET_FUSED_ABLATE=64 (timing only)
    if abl:
        os.environ["SONET_FUSED_ABLATE"] = abl
    _lib.LIB_PATH = os.path.join(ROOT, "so-net_amd", "lib", "variants", "libsonet_hip_%s.so" % libname)
    from sonet_hip import ops, synth
    from models import layers as Lm
    import numpy as np
    dev = torch.device("cuda:0")
    B = int(os.environ.get("B", "64"))
    pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.to(dev).eval()
    inp = synth.make_inputs(B, 5000, seed=1, device=dev)
    a = ops.som_assign(inp["pc"], inp["node"], 3)
    sg = ops.som_sort_group(inp["pc"], inp["sn"], a)
    g = ops.som_group(inp["pc"], inp["sn"], a, want_augmented=True)
    wstream, affine = pr._fused_state()
    res = {}
    for mode, run in (("pool", lambda: ops.pointresnet_fused_pool(sg, wstream, affine, 64)),
                      ("store", lambda: ops.pointresnet_fused(g["x_augmented"], wstream, affine))):
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(int(os.environ.get("ITERS", "15"))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[mode] = out[:, :, ::97].contiguous().cpu().numpy() if mode == "store" else out.cpu().numpy()
        print("%-14s %-5s median %.4f ms  min %.4f ms (incl. init/decode launches for pool)" % (name, mode, ts[len(ts) // 2], ts[0]), flush=True)
    np.savez(out_path, **res)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    import numpy as np
    outdir = "/tmp/sonet_variants"                               # (scratch: the store-path outputs are large)
    os.makedirs(outdir, exist_ok=True)
    first = None
    for name in sys.argv[1:]:
        path = os.path.join(outdir, name + ".npz")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, path], timeout=240)
        except subprocess.TimeoutExpired:
            print("%-14s TIMEOUT (killed)" % name, flush=True)
            continue
        if r.returncode != 0:
            print("%-14s FAILED rc=%d" % (name, r.returncode), flush=True)
            continue
        cur = np.load(path)
        if "@" in name:
            continue
        if first is None:
            first = cur
        else:
            for k in first.files:
                same = np.array_equal(first[k].view(np.uint32), cur[k].view(np.uint32))
                print("%-14s %-5s output %s the first variant's" % (name, k, "bit-identical to" if same else "DIFFERS from"), flush=True)
                if not same:
                    a64, b64 = first[k].astype(np.float64), cur[k].astype(np.float64)
                    print("%-14s %-5s   max |diff| %.3g, rms diff / rms value %.3g, %d of %d elements differ" % (
                        name, k, np.abs(a64 - b64).max(), np.sqrt(np.mean((a64 - b64) ** 2)) / np.sqrt(np.mean(a64 ** 2)),
                        int((first[k].view(np.uint32) != cur[k].view(np.uint32)).sum()), a64.size), flush=True)
