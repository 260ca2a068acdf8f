"""tools/pmc_traffic.py -- turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into
profiles/pmc_traffic.json: HBM bytes per launch per kernel (bench.py's roofline.traffic).

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 bytes here; on gfx950 FETCH_SIZE counts wide (16 B/lane) coalesced streaming reads at HALF their size, so the
read side of kernels that stream with 16-byte loads (index_max) is doubled; dword-granular readers are left as is.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDE_READERS = ("index_max",)
NAMES = [("pointresnet_bf16_pool2_kernel", "pointresnet_bf16_pool_L15000"), ("pointresnet_fused_kernel", "pointresnet_fused_pool_L15000"), ("index_max_kernel", "index_max_gather"),
         ("som_assign_rank_kernel", "som_assign_sort"), ("som_sort_fill2_kernel", "som_assign_sort"), ("som_assign_keys_kernel", "som_assign"), ("som_assign_kernel", "som_assign"), ("som_sort_group_kernel", "som_sort_group"), ("som_group_kernel", "som_group")]


def load(counter):
    path = os.path.join(ROOT, "gpurun_out", "pmc_traffic_%s" % counter, "pmc_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "B64_N5000"
    fetch, write = load("FETCH_SIZE"), load("WRITE_SIZE")
    out = {}
    for needle, name in NAMES:
        f = next((v for k, v in fetch.items() if needle in k), None)
        w = next((v for k, v in write.items() if needle in k), None)
        if f is None or w is None:
            continue
        fbytes = f * 1024 * (2 if name.startswith(WIDE_READERS) else 1)
        out[name] = out.get(name, 0) + int(fbytes + w * 1024)     # (several kernels of one entry point add up)
        print("%-28s %-28s FETCH_SIZE %.4g  WRITE_SIZE %.4g  -> %.1f MB per launch" % (name, needle, f, w, (fbytes + w * 1024) / 1e6))
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    allv = json.load(open(path)) if os.path.exists(path) else {}
    allv[tag] = out
    json.dump(allv, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
