"""What runs between two launches of the fused first PointNet when P graphs are replayed round-robin (the headline of bench.py)?

Input: the kernel trace of `rocprofv3 --kernel-trace --output-format csv -- python bench.py ...` (tools/gpu_r6g.sh).  The fused kernel owns
every CU while it runs, so the step time is (fused kernel) + (gap); this prints the distribution of the gaps, how much of a gap is covered
by at least one other kernel, the kernels by their total time inside gaps, and the steady-state period."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
fused = [r for r in rows if "pointresnet_fused_kernel" in r[2]]
print("kernels %d, fused launches %d" % (len(rows), len(fused)))
# the P-graphs-in-flight region: the longest run of fused launches in which consecutive launches come from different queues
best, cur = (0, 0), 0
for i in range(1, len(fused) + 1):
    if i == len(fused) or fused[i][3] == fused[i - 1][3]:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = i
fused = fused[best[0] + 10:best[1] - 5]
print("round-robin region: %d fused launches on queues %s" % (len(fused), sorted({r[3] for r in fused})))
t_lo, t_hi = fused[0][0], fused[-1][1]
per = [(b[0] - a[0]) / 1e3 for a, b in zip(fused, fused[1:])]
dur = [(r[1] - r[0]) / 1e3 for r in fused]
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(fused, fused[1:])]


def stat(v):
    v = sorted(v)
    return "mean %.1f  p10 %.1f  median %.1f  p90 %.1f" % (sum(v) / len(v), v[len(v) // 10], v[len(v) // 2], v[9 * len(v) // 10])


print("fused duration us: " + stat(dur))
print("fused start-to-start us: " + stat(per))
print("gap (end -> next start) us: " + stat(gaps))
ov = [max(0, min(a[1], b[1]) - max(a[0], b[0])) for a, b in zip(fused, fused[1:])]
print("two fused launches overlapping: %d of %d pairs" % (sum(1 for o in ov if o > 0), len(ov)))
others = [r for r in rows if "pointresnet_fused_kernel" not in r[2] and r[1] > t_lo and r[0] < t_hi]
# time of each other kernel inside gaps / under a fused launch
in_gap, under = defaultdict(float), defaultdict(float)
cnt = defaultdict(int)
fi = 0
for s, e, name, q, st in others:
    cnt[name] += 1
    tot = e - s
    u = 0
    for fs, fe, *_ in fused:
        if fe <= s:
            continue
        if fs >= e:
            break
        u += max(0, min(e, fe) - max(s, fs))
    under[name] += u / 1e3
    in_gap[name] += (tot - u) / 1e3
n = len(fused)
print("per fused launch, other kernels: in gaps %.1f us, under a fused launch %.1f us" % (sum(in_gap.values()) / n, sum(under.values()) / n))
for name in sorted(in_gap, key=lambda k: -(in_gap[k] + under[k]))[:24]:
    print("  %-72s x%5.2f  in-gap %6.1f us  under-fused %6.1f us  mean dur %6.1f us" % (
        name[:72], cnt[name] / n, in_gap[name] / n, under[name] / n, (in_gap[name] + under[name]) / max(1, cnt[name])))
# coverage of the gaps: fraction of gap time with >= 1 / >= 2 kernels running
ev = []
for s, e, *_ in others:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
cov1 = cov2 = idle = 0
gi = 0
gap_iv = [(a[1], b[0]) for a, b in zip(fused, fused[1:]) if b[0] > a[1]]
depth, last = 0, None
for t, d in ev:
    if last is not None and t > last:
        # add [last, t) at `depth` restricted to gaps
        for gs, ge in gap_iv:
            if ge <= last:
                continue
            if gs >= t:
                break
            o = min(t, ge) - max(last, gs)
            if o > 0:
                if depth >= 1:
                    cov1 += o
                if depth >= 2:
                    cov2 += o
    depth += d
    last = t
gt = sum(ge - gs for gs, ge in gap_iv)
print("gap time %.1f us per launch: >=1 kernel running %.0f %%, >=2 running %.0f %%, nothing running %.0f %%" % (gt / 1e3 / n, 100 * cov1 / gt, 100 * cov2 / gt, 100 * (gt - cov1) / gt))
