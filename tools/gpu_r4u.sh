#!/bin/bash
# HBM / L2 counters of the bf16 streaming layer kernel on 320 -> 384, rows of 15000 (30000 B: 16 B off the 32-byte sectors) vs 15040 columns
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04u; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
pmc() { tag=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o pmc -- python $R/tools/one_bf16.py 64 256 384 $L 3 > /dev/null 2> $O/pmc_$tag.err; }
for L in 15000 15040; do
  export L
  pmc ${L}_f FETCH_SIZE
  pmc ${L}_w WRITE_SIZE
  pmc ${L}_c TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  pmc ${L}_d TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_WRITE_sum
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cd $R; python - <<'PY'
import csv, glob, os
for d in sorted(glob.glob('gpurun_out/r04u/pmc_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if 'bf16r' in r['Kernel_Name'] or 'pointmlp_bf16' in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        print(os.path.basename(d), {k: (sum(v) / len(v), len(v)) for k, v in acc.items()})
PY
