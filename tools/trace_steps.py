"""tools/trace_steps.py TRACE.csv -- from a rocprofv3 --kernel-trace csv: the timeline (start offset, duration, gap to the previous kernel)
of one steady-state step per distinct step shape (a step = the launches from one som_assign_rank_kernel to the next)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    if "som_assign_rank" in r["Kernel_Name"] and cur:
        steps.append(cur)
        cur = []
    cur.append(r)
seen = {}
for st in steps[len(steps) // 2:]:
    key = tuple(r["Kernel_Name"][:60] for r in st)
    if key in seen or len(st) < 8:
        continue
    seen[key] = True
    t0 = int(st[0]["Start_Timestamp"])
    prev_end = t0
    print("---- step with %d launches, %.1f us from first start to last end" % (len(st), (int(st[-1]["End_Timestamp"]) - t0) / 1e3))
    for r in st:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        print("  +%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
        prev_end = e
    if len(seen) >= 3:
        break
