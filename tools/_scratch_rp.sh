#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/rp_$v
  ITERS=31 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$v -o r -- python $GRAFT_REPO_ROOT/tools/fused_variants.py --child $v /tmp/$v.npz > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/rp_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; if [ -n "$f" ]; then python3 -c "
import csv
for i,r in enumerate(csv.DictReader(open('$f'))):
    if i<5: print('%-60s calls %4s avg %10.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
"; fi
done
python $GRAFT_REPO_ROOT/tools/fused_variants.py dbase dnew 2>&1 | grep -E "identical|DIFF"
