"""tools/check_bf16r_asm.py [ASM] -- audit of the streaming bf16 layer kernel (pointmlp_bf16r_kernel, pointmlp_bf16.hip).

Its X loads are inline asm that hipcc cannot see as memory operations: a register of the X ring holds garbage from the moment its
load is issued until the hand-counted s_waitcnt in front of the consumer.  hipcc is free to MOVE such a register (a copy for a tied
asm operand, a phi copy on a loop edge) -- and a move before the wait reads the stale value.  This script compiles the file to
assembly (or reads ASM) and fails if, in any instantiation, an instruction outside the asm statements reads or writes a register
that is in flight, a wait+perm statement reads a register that no load was issued into, or the kernel spills."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(operands):
    regs = re.findall(r'\bv(\d+)\b', operands)
    rng = re.findall(r'v\[(\d+):(\d+)\]', operands)
    return {'v' + r for r in regs} | {'v%d' % k for a, b in rng for k in range(int(a), int(b) + 1)}


def audit(txt):
    """Linear scan per kernel: a register is IN FLIGHT from the asm buffer_load_dword that names it until an asm statement holding an
    s_waitcnt reads it (the wait + perm statement) or the closing ``s_waitcnt vmcnt(0)``.  Any instruction outside the asm statements
    that reads or writes an in-flight register is a finding (the text order stands in for the control flow: the loop bodies are
    straight-line and end with the same registers in flight as they start with)."""
    parts = re.split(r'\n(_Z[\w]+):', txt)
    found, problems = 0, []
    for i in range(1, len(parts), 2):
        if 'pointmlp_bf16r_kernel' not in parts[i]:
            continue
        found += 1
        name = parts[i][-34:]
        body = parts[i + 1].split('.end_amdhsa_kernel')[0]
        lines = [l.strip() for l in body.split('\n')]
        inflight, loads, waits = set(), 0, 0
        n = 0
        while n < len(lines):
            l = lines[n]
            if l.startswith(';;#ASMSTART'):
                block = []
                n += 1
                while not lines[n].startswith(';;#ASMEND'):
                    block.append(lines[n])
                    n += 1
                if any(x.startswith('s_waitcnt vmcnt(0)') for x in block):
                    inflight.clear()
                elif any(x.startswith('s_waitcnt') for x in block):
                    waits += 1
                    srcs = set()
                    for x in block:
                        if x.startswith('v_perm_b32'):
                            srcs |= regs_of(x.split(None, 1)[1].split(',', 1)[1])
                    if not srcs <= inflight:
                        problems.append('%s: line %d: a wait+perm statement reads %s, not in flight' % (name, n, sorted(srcs - inflight)))
                    inflight -= srcs
                else:
                    for x in block:
                        m = re.match(r'buffer_load_dword (v\d+),', x)
                        if m:
                            loads += 1
                            if m.group(1) in inflight:
                                problems.append('%s: line %d: load into %s, still in flight' % (name, n, m.group(1)))
                            inflight.add(m.group(1))
            elif l and not l.startswith((';', '.')) and inflight:
                ops = l.split(None, 1)
                if len(ops) == 2 and regs_of(ops[1]) & inflight:
                    problems.append('%s: line %d touches in-flight %s: %s' % (name, n, sorted(regs_of(ops[1]) & inflight), l))
            n += 1
        mt = int(re.search(r'pointmlp_bf16r_kernelILi(\d+)E', parts[i]).group(1))
        pool = re.search(r'pointmlp_bf16r_kernelILi\d+ELb[01]ELb1E', parts[i]) is not None
        nst = sum(1 for l in lines if l.startswith('buffer_store_dword'))
        # the wait counts assume 16 MT stores per epilogue and as many in the prologue -- or (the pooling variant, which writes its
        # result after the closing vmcnt(0)) NO vector-memory operation but the X loads anywhere in the loop
        want = 0 if pool else 32 * mt
        if nst != want:
            problems.append('%s: %d buffer_store_dword (expected %d)' % (name, nst, want))
        if pool:
            # (the loop ends at the kernel's OWN closing wait -- the asm statement; what hipcc emits behind it is the result's write-out)
            last0 = max((k for k, l in enumerate(lines) if l.startswith('s_waitcnt vmcnt(0)') and k > 0 and lines[k - 1].startswith(';;#ASMSTART')), default=-1)
            # (vector-memory instructions hipcc emitted itself inside the loop would sit between asm blocks: count them by position)
            inasm, vm_in_loop, seen_first_load = False, 0, False
            for l in lines[:last0 if last0 >= 0 else len(lines)]:
                if l.startswith(';;#ASMSTART'):
                    inasm = True
                elif l.startswith(';;#ASMEND'):
                    inasm = False
                elif inasm and l.startswith('buffer_load_dword'):
                    seen_first_load = True
                elif not inasm and seen_first_load and l.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                    vm_in_loop += 1
            if vm_in_loop:
                problems.append('%s: %d compiler-emitted vector-memory instructions inside the loop of the pooling variant' % (name, vm_in_loop))
        if loads < 64 or waits < 8:
            problems.append('%s: %d asm loads / %d waits found (the kernel changed shape?)' % (name, loads, waits))
        m = re.search(r'; ScratchSize: (\d+)', parts[i + 1])
        if m and int(m.group(1)) != 0:
            problems.append('%s: scratch %s bytes' % (name, m.group(1)))
    return found, problems


def main():
    if len(sys.argv) > 1:
        txt = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, 'bf16.s')
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                                   '-ffp-contract=off', '-fno-fast-math', '-I' + os.path.join(ROOT, 'include'),
                                   os.path.join(ROOT, 'so-net_amd', 'csrc', 'pointmlp_bf16.hip'), '-o', out],
                                  stderr=subprocess.DEVNULL)
            txt = open(out).read()
    found, problems = audit(txt)
    if found != 12:
        problems.append('%d instantiations of pointmlp_bf16r_kernel (expected 12: four storing, three pooling, two storing + three pooling with normalise-on-load)' % found)
    for p in problems[:20]:
        print('PROBLEM', p)
    print('check_bf16r_asm: %d kernels, %d problems' % (found, len(problems)))
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
