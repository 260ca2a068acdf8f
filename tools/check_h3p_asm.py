"""tools/check_h3p_asm.py -- audit of what hipcc made of pointmlp_h3p.hip (run by __graft_entry__.build() and the CPU tests).

Every vector-memory LOAD of the layer kernels' pass loop is inline asm with hand-counted waits; the build is valid only if hipcc added no
vmcnt wait of its own between the first MFMA and the last store of a kernel (a spill reload or a compiler-visible load inside the loop
would make it wait for vmcnt(0) there and drain the look-ahead every iteration), and if no kernel keeps scratch accesses inside that span."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "so-net_amd", "csrc", "pointmlp_h3p.hip")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function".split()


def main(extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["--cuda-device-only", "-S", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    parts = re.split(r'\n(_Z[\w]+):', txt)
    bad = 0
    n_k = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        if "pointmlp_h3p_kernel" not in name:
            continue
        n_k += 1
        lines = body.split(".end_amdhsa_kernel")[0].split("\n")
        mf = [n for n, l in enumerate(lines) if "v_mfma" in l]
        st = [n for n, l in enumerate(lines) if re.search(r"buffer_store_dword", l)]
        lo, hi = mf[0], max(mf[-1], st[-1] if st else 0)
        inasm, waits, scratch = False, [], []
        for n, l in enumerate(lines):
            s = l.strip()
            if "ASMSTART" in s:
                inasm = True
            elif "ASMEND" in s:
                inasm = False
            if lo <= n <= hi and not inasm:
                if s.startswith("s_waitcnt") and "vmcnt" in s:
                    waits.append((n, s))
                if s.startswith("scratch_"):
                    scratch.append((n, s))
        short = re.search(r"kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name).groups()
        if waits or scratch:
            bad += 1
            print("FAIL %s: compiler vmcnt waits %s, scratch %s" % (",".join(short), waits[:3], scratch[:3]))
    print("%s  %d layer kernels audited, %d with compiler waits / scratch inside the pass loop" % ("ok" if not bad else "FAIL", n_k, bad))
    return 1 if bad or not n_k else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
