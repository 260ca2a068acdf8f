"""tools/pmc_sq_summary.py <kernel substring> <csv> [<csv> ...] -- mean per launch of every counter in rocprofv3 --pmc CSVs."""
import csv
import sys
from collections import defaultdict

name = sys.argv[1]
acc, cnt = defaultdict(float), defaultdict(int)
for path in sys.argv[2:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            if name in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
m = {k: acc[k] / cnt[k] for k in acc}
for k in sorted(m):
    print("%-28s %16.0f   (%d launches)" % (k, m[k], cnt[k]))
g = m.get
if g("SQ_INSTS_MFMA"):
    print("VALU per MFMA               %.2f" % ((g("SQ_INSTS_VALU", 0) - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")))
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if g(k) is not None:
            print("%-28s %.3f of wave cycles" % (k, g(k) / g("SQ_WAVE_CYCLES")))
if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
    print("MFMA busy / SQ busy         %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
