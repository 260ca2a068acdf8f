"""tools/check_fused_asm.py [SOURCE] [-D...] -- audit of a build of the fused first-PointNet kernel whose accumulation registers are
owned by inline asm (literal a[N:M] operands the compiler does not know about).

The build is valid only if hipcc itself never touches the accumulation file: a compiler-generated v_accvgpr_* (a VGPR spilled into
what it believes is a free AGPR) would be overwritten by the kernel's MFMAs, silently.  Checks, per kernel of the source:
  * no scratch (VGPR spill to memory), * no v_accvgpr_* outside an ASMSTART/ASMEND region, * MFMA count (information).
Exit code 1 if any check fails.  Runs on the CPU box (hipcc -S for gfx950)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function".split()


def audit(src, extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur, inasm = {}, None, False
    for ln in text:
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            cur = kernels.setdefault(m.group(1), dict(mfma=0, accvgpr_by_compiler=0, scratch=0))
            inasm = False
            continue
        if cur is None:
            continue
        if "ASMSTART" in s:
            inasm = True
        elif "ASMEND" in s:
            inasm = False
        elif s.startswith("v_mfma"):
            cur["mfma"] += 1
        elif s.startswith("v_accvgpr") and not inasm:
            cur["accvgpr_by_compiler"] += 1
        elif s.startswith("scratch_"):
            cur["scratch"] += 1
    return kernels


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = [a for a in args if a.startswith("-")]
    srcs = [a for a in args if not a.startswith("-")] or [os.path.join(ROOT, "so-net_amd", "csrc", "pointresnet_fused.hip")]
    bad = False
    for src in srcs:
        for name, k in audit(src, extra).items():
            if "fused_kernel" not in name:
                continue
            ok = k["accvgpr_by_compiler"] == 0 and k["scratch"] == 0
            bad |= not ok
            print("%s %-40s mfma %d  compiler v_accvgpr %d  scratch %d" % ("ok " if ok else "BAD", name[14:54], k["mfma"], k["accvgpr_by_compiler"], k["scratch"]))
    sys.exit(1 if bad else 0)
