"""tools/isa_summary.py ASM [KERNEL_SUBSTRING] -- per-kernel register / scratch / LDS summary of a hipcc -S output, and (with a
kernel substring) a compressed listing of the basic block that holds the most MFMAs (the main loop)."""
import re
import sys


def kernels(txt):
    parts = re.split(r'\n(_Z[\w]+):', txt)
    for i in range(1, len(parts), 2):
        yield parts[i], parts[i + 1]


def meta(body, k):
    m = re.search(r'; %s: (\S+)' % k, body)
    return m.group(1) if m else '?'


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else None
    for name, body in kernels(txt):
        if 'kernel' not in name:
            continue
        code = body.split('.end_amdhsa_kernel')[0]
        short = re.sub(r'^_ZN\d+_GLOBAL__N_1', '', name)[:70]
        print('%-72s vgpr %s agpr %s scratch %s occ %s lds %s mfma %d' % (short, meta(body, 'NumVgprs'), meta(body, 'NumAgprs'), meta(body, 'ScratchSize'),
                                                                   meta(body, 'Occupancy'), meta(body, 'LDSByteSize'), len(re.findall(r'v_mfma', code))))
        if want and want in name:
            lines = body.split('\n')
            hdrs = [n for n, l in enumerate(lines) if l.strip().startswith('.LBB')]
            best = None
            for a, b in zip(hdrs, hdrs[1:] + [len(lines)]):
                c = sum('v_mfma' in x for x in lines[a:b])
                if best is None or c > best[0]:
                    best = (c, a, b)
            _, a, b = best
            res, prev, k = [], None, 0
            for l in lines[a:b]:
                s = l.strip()
                if not s or s.startswith(';'):
                    continue
                op = s.split()[0]
                key = s if op.startswith(('s_waitcnt', 's_nop', 's_barrier')) else op
                if key == prev:
                    k += 1
                else:
                    if prev:
                        res.append('%s x%d' % (prev, k))
                    prev, k = key, 1
            res.append('%s x%d' % (prev, k))
            print('\n'.join(res))


if __name__ == '__main__':
    main()
