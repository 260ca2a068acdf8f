"""tools/check_fused_asm.py [SOURCE ...] [-D...] -- audit of the builds of the kernels whose accumulation registers are owned by
inline asm (literal a[N:M] operands the compiler does not know about): the fused first PointNet (pointresnet_fused.hip).

The build is valid only if hipcc itself never touches the accumulation file: a compiler-generated v_accvgpr_* (a VGPR spilled into
what it believes is a free AGPR) would be overwritten by the kernel's MFMAs, silently.  Checks, per kernel of the source:
  * no scratch (VGPR spill to memory), * no v_accvgpr_* outside an ASMSTART/ASMEND region,
  * no instruction other than an MFMA touches the VGPR destination of an asm MFMA before two later MFMAs or an `s_nop 15` have been
    issued (hipcc takes an asm statement's outputs for available at once and pads nothing: a copy or a VALU use placed right behind
    an MFMA statement would read the registers before the matrix pipe has written them),
  * MFMA count (information).
Exit code 1 if any check fails.  Runs on the CPU box (hipcc -S for gfx950)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM_OWNED = ["pointresnet_fused.hip"]                          # sources with kernels that own a[...] through inline asm
ASM_OWNED_KERNELS = ["pointresnet_fused_kernel"]
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function".split()


def audit(src, extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur, inasm = {}, None, False
    pending = []                                                # [set of VGPR numbers an MFMA is writing, MFMAs issued since]
    for ln in text:
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            cur = kernels.setdefault(m.group(1), dict(mfma=0, accvgpr_by_compiler=0, scratch=0, early_reads=0))
            inasm, pending = False, []
            continue
        if cur is None:
            continue
        if "ASMSTART" in s:
            inasm = True
        elif "ASMEND" in s:
            inasm = False
        elif s.startswith("v_mfma"):
            cur["mfma"] += 1
            for pnd in pending:
                pnd[1] += 1
            pending = [pnd for pnd in pending if pnd[1] < 2]
            d = re.match(r"v_mfma\S*\s+v\[(\d+):(\d+)\]", s)
            if d:
                pending.append([set(range(int(d.group(1)), int(d.group(2)) + 1)), 0])
            continue
        elif s.startswith("v_accvgpr") and not inasm:
            cur["accvgpr_by_compiler"] += 1
        elif s.startswith("scratch_"):
            cur["scratch"] += 1
        if s.startswith("s_nop 15"):
            pending = []
        elif pending and not s.startswith((";", ".", "s_")) and re.match(r"^[a-z]", s):
            ops = s.split(None, 1)
            srcs = ops[1].split(",") if len(ops) > 1 else []          # every operand: a write into a pending destination is as wrong
            used = set()
            for o in srcs:
                for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", o):
                    used.update(range(int(a), int(b) + 1))
                used.update(int(a) for a in re.findall(r"\bv(\d+)\b", o))
            if any(used & pnd[0] for pnd in pending):
                cur["early_reads"] += 1
    return kernels


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = [a for a in args if a.startswith("-")]
    srcs = [a for a in args if not a.startswith("-")] or [os.path.join(ROOT, "so-net_amd", "csrc", f) for f in ASM_OWNED]
    bad = False
    for src in srcs:
        for name, k in audit(src, extra).items():
            if not any(k_ in name for k_ in ASM_OWNED_KERNELS):
                continue
            ok = k["accvgpr_by_compiler"] == 0 and k["scratch"] == 0 and k["early_reads"] == 0
            bad |= not ok
            print("%s %-40s mfma %d  compiler v_accvgpr %d  scratch %d  early reads of MFMA results %d"
                  % ("ok " if ok else "BAD", name[14:54], k["mfma"], k["accvgpr_by_compiler"], k["scratch"], k["early_reads"]))
    sys.exit(1 if bad else 0)
