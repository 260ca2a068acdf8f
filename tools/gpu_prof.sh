#!/bin/bash
# rocprofv3 kernel stats of a bench.py command line:  tools/gpu_prof.sh TAG -- <bench args>
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=/tmp/prof_$TAG; mkdir -p $O $R/gpurun_out      # (raw traces stay off the 64 MiB return path)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python $R/bench.py "$@" --no-cpu-baseline > $O/stdout.log 2> $O/stderr.log
f=$(find $O -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats.csv && head -25 $f | cut -c1-160
