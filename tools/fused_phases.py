"""tools/fused_phases.py [pool|store] -- per-phase shader cycles of the fused first-PointNet kernel, from the
profiling build (make -C so-net_amd/csrc prof).  Prints mean cycles per tile per wave next to the MFMA-only ideal."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd"))
import torch  # noqa: E402
from sonet_hip import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "so-net_amd", "lib", "libsonet_hip_prof.so")
if os.environ.get("VARIANT"):                                  # a tools/build_variant.sh build with -DSONET_PROF
    _lib.LIB_PATH = os.path.join(ROOT, "so-net_amd", "lib", "variants", "libsonet_hip_%s.so" % os.environ["VARIANT"])
from sonet_hip import ops, synth  # noqa: E402
from models import layers as Lm  # noqa: E402
import numpy as np  # noqa: E402

DEV = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "pool"
B = int(os.environ.get("B", "64"))
pr = Lm.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
synth.fill_state_dict_(pr.state_dict(), seed=7)
pr.to(DEV).eval()
inp = synth.make_inputs(B, 5000, seed=1, device=DEV)
a = ops.som_assign(inp["pc"], inp["node"], 3)
wstream, affine = pr._fused_state()
if mode == "pool":
    sg = ops.som_sort_group(inp["pc"], inp["sn"], a)
    run = lambda: ops.pointresnet_fused_pool(sg, wstream, affine, 64)
else:
    g = ops.som_group(inp["pc"], inp["sn"], a, want_augmented=True)
    run = lambda: ops.pointresnet_fused(g["x_augmented"], wstream, affine)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib = _lib.load()
lib.sonet_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
NP = 32
buf = np.zeros(1024 * NP, dtype=np.int64)
assert lib.sonet_prof_read(buf.ctypes.data, buf.size) == 0
p = buf.reshape(1024, NP).astype(np.float64)
GEN4 = os.environ.get("GEN", "4") == "4"
TP = 256 if GEN4 else 64
tiles = B * ((15000 + TP - 1) // TP) / float(min(256, int(os.environ.get("SONET_FUSED_MAXCU", "256"))))
if GEN4:
    # phase slots of the fourth-generation kernel (PROF_MARK in pointresnet_fused.hip); MFMA-only ideal = 32 cycles per MFMA
    slots = [(0, "kernel prologue", 0), (1, "tile prologue (split x)", 0), (6, "layer 1 + first jobs", 12), (7, "layer 2 + transition jobs", 96),
             (8, "layer 3, tiles 0-3", 192), (9, "layer 3, tiles 4-7", 192), (2, "layer 1 again (exposed)", 12), (10, "layer 4, pass 0 (+ jobs)", 360), (3, "layer 4, passes 1-3", 1080),
             (4, "epilogues (4)", 0), (5, "tail", 0)]
    waits = [(28, "  of which: wait for own W pieces (vmcnt)"), (29, "  of which: stage barriers")]
else:
    slots = [(0, "kernel prologue", 0), (1, "first front (exposed)", 0), (2, "tile prologue + barrier 1", 0), (3, "layer 3", 96),
             (4, "layer 4 steps 0-3 + layer-3 jobs", 72), (5, "barrier 2", 0), (6, "layer 4 steps 4-19 + next front", 324), (7, "epilogue", 0), (8, "tail", 0)]
    waits = []
idx = [i for i, _, _ in slots]
tot = p[:, idx].sum(1).mean()
print("mode %s: %.0f cycles per wave, %.2f tiles per workgroup, MFMA-only ideal per tile %d" % (mode, tot, tiles, 32 * sum(m for _, _, m in slots)))
for i, n, m in slots + [(i, n, 0) for i, n in waits]:
    per_tile = p[:, i].mean() / tiles
    print("  %-44s %9.0f cycles/tile  (%4.1f%%)%s" % (n, per_tile, 100 * p[:, i].mean() / tot,
          "   MFMA-only ideal %d -> %.0f%%" % (32 * m, 100 * 32 * m / per_tile) if m else ""))
pw = p.reshape(-1, 4, NP)
print("  per wave (cycles/tile): total " + " ".join("%8s" % ("slot%d" % i) for i in idx))
for w in range(4):
    m = pw[:, w, :].mean(0) / tiles
    print("    wave %d           %8.0f " % (w, pw[:, w, idx].sum(1).mean() / tiles) + " ".join("%8.0f" % m[i] for i in idx))
rt = p[:, 31].mean()                                            # s_memrealtime ticks (100 MHz) per wave
cyc = tot
print("  shader clock while the kernel runs: %.0f cycles in %.1f us = %.3f GHz" % (cyc, rt / 100.0, cyc / (rt * 10.0) if rt else 0.0))
