"""tools/check_dma_hazard.py -- scan the built gfx950 code objects for the hazard that cost round 2 a day:

    a VMEM instruction (here: the hand-written LDS-DMA, global_load_lds_* / buffer_load_* ... lds) that reads an SGPR written by a
    VALU instruction (v_readlane_b32 of a spilled SGPR, v_readfirstlane_b32, v_cmp into an SGPR pair) needs 5 wait states in between.
    hipcc pads its own instructions, NOT the contents of an inline-asm statement, so a pointer the compiler restores right in
    front of the asm reaches the DMA stale (deterministic garbage, no fault).

Usage: python tools/check_dma_hazard.py [object files ...]   (default: so-net_amd/build/*.o); exit code 1 when a candidate is found.
Also imported by tests/test_isa_hazards.py."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32")
NEED = 5


def disassemble(obj):
    """-> list of (mnemonic, operand string) of every gfx950 instruction in a host object with an embedded HIP fat binary."""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", local], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        cos = [p for p in glob.glob(local + ".*") if "amdgcn" in p]
        out = []
        for co in cos:
            txt = subprocess.run([OBJDUMP, "-d", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode("utf-8", "replace")
            for line in txt.splitlines():
                m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
                if m:
                    out.append((m.group(1), m.group(2)))
        return out


def sgprs(text):
    regs = set()
    for lo, hi in re.findall(r"s\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    for r in re.findall(r"(?<![a-z\[:])s(\d+)\b", text):
        regs.add(int(r))
    return regs


def scan(instrs):
    """-> list of (index, writer, consumer) hazard candidates."""
    found = []
    for i, (mn, ops) in enumerate(instrs):
        if not (mn.startswith("global_load_lds") or (mn.startswith("buffer_load") and " lds" in " " + ops)):
            continue
        need = sgprs(ops)
        states, k = 0, i - 1
        while k >= 0 and states < NEED:
            pm, po = instrs[k]
            if pm in VALU_SGPR_WRITERS:
                dst = sgprs(po.split(",")[0])
                if dst & need:
                    found.append((i, "%s %s" % (pm, po), "%s %s" % (mn, ops)))
                    break
            states += (int(po.strip() or 0) + 1) if pm == "s_nop" else 1
            k -= 1
    return found


def main(argv):
    objs = argv or sorted(glob.glob(os.path.join(ROOT, "so-net_amd", "build", "*.o")))
    bad = 0
    for o in objs:
        ins = disassemble(o)
        n_dma = sum(1 for mn, ops in ins if mn.startswith("global_load_lds") or (mn.startswith("buffer_load") and " lds" in " " + ops))
        hz = scan(ins)
        print("%-28s %7d instructions, %4d LDS-DMA, %d hazard candidate(s)" % (os.path.basename(o), len(ins), n_dma, len(hz)))
        for _, w, c in hz[:5]:
            print("    %s   ->   %s" % (w, c))
        bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
