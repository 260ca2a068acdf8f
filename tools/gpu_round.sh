#!/bin/bash
# tools/gpu_round.sh TAG  -- one gpurun call that produces everything a round's measurement section needs
# (run ON the GPU box:  gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a').  Writes under gpurun_out/TAG/ and
# copies the judged summaries into profiles/ (the copies come back only through gpurun_out/: re-copy locally with
#   cp gpurun_out/TAG/profiles/* profiles/   after the call).
#   1. pytest -m gpu                     -> TAG_pytest_gpu.log
#   2. smoke()                           -> TAG_smoke.log
#   3. bench.py (forward, the metric)    -> TAG_bench_forward.json
#   4. rocprofv3 --kernel-trace --stats of the same command -> TAG_kernel_stats_forward.csv
#   5. PMC FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) -> pmc_traffic.json
#   6. bench.py --mode train             -> TAG_bench_train.json
#   7. stand-alone kernel microbenchmarks (index_max / som rooflines) -> TAG_microbench.log
# Steps are independent: a failing one is reported and the rest still run.  SKIP="1 5" skips steps.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; P=$O/profiles
mkdir -p $P
cd $R
skip() { [[ " $SKIP " == *" $1 "* ]]; }
run() { echo "== step $1: $2"; }
export TMPDIR=/tmp

skip 1 || { run 1 "pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > $P/${TAG}_pytest_gpu.log 2>&1; tail -3 $P/${TAG}_pytest_gpu.log; }
skip 2 || { run 2 smoke; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $P/${TAG}_smoke.log 2>&1; tail -1 $P/${TAG}_smoke.log; }
skip 3 || { run 3 "bench forward"; timeout 600 python bench.py --steps 50 --warmup 10 2> $O/bench_forward.err | tail -1 > $P/${TAG}_bench_forward.json; head -c 400 $P/${TAG}_bench_forward.json; echo; }
skip 4 || { run 4 "rocprofv3 kernel stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_fwd -o fwd -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-other-configs > /dev/null 2> $O/rocprof_fwd.err);
            f=$(find /tmp/rp_${TAG}_fwd -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_forward.csv && head -8 $f; }
skip 5 || { run 5 "PMC traffic"; timeout 900 bash tools/pmc_traffic.sh; python tools/pmc_traffic.py B64_N5000 && cp profiles/pmc_traffic.json $P/pmc_traffic.json; rm -rf $R/gpurun_out/pmc_traffic_FETCH_SIZE $R/gpurun_out/pmc_traffic_WRITE_SIZE; }   # (raw traces stay off the 64 MiB return path)
skip 6 || { run 6 "bench train"; timeout 600 python bench.py --mode train --steps 20 --warmup 5 2> $O/bench_train.err | tail -1 > $P/${TAG}_bench_train.json; head -c 300 $P/${TAG}_bench_train.json; echo; }
skip 7 || { run 7 microbench; timeout 600 python tools/microbench.py > $P/${TAG}_microbench.log 2>&1; grep -i "index_max\|som " $P/${TAG}_microbench.log | head; }
echo "== done: $(ls $P | wc -l) files under gpurun_out/$TAG/profiles"
