#!/bin/bash
# tools/ab_tree.sh -- time an EARLIER commit's tree beside the current one on ONE GPU box (docs/findings.md R6.5: boxes differ by +-3 %, a
# regression of that size hides in the spread unless both trees run in the same gpurun call).
#
#   here:        bash tools/ab_tree.sh prepare <commit>        # git archive into _ab_<commit>/ (add it to .gitignore), build its library + oracle
#   on the box:  gpurun --timeout 1200 -- 'bash tools/ab_tree.sh run _ab_<commit> "--mode train" 3'
#                # <tree> "<bench.py flags>" <repetitions>: alternates old / new, prints ms_per_step and value of each run
#   afterwards:  rm -rf _ab_<commit>   (it travels with every gpurun snapshot while it exists)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
prepare)
    c=$2; d=$R/_ab_$c
    rm -rf "$d"; mkdir -p "$d"
    git -C "$R" archive "$c" | tar -x -C "$d"
    grep -qx "_ab_$c/" "$R/.gitignore" || echo "_ab_$c/" >> "$R/.gitignore"
    make -j8 -C "$d/so-net_amd/csrc" all > "$d/build.log" 2>&1 || { tail -20 "$d/build.log"; exit 1; }
    (cd "$d" && python -c "from oracle import build_ref; build_ref.build_oracle()" > /dev/null 2>&1 || true)
    ls -la "$d/so-net_amd/lib/"
    ;;
run)
    tree=$2; flags=${3:-"--mode train"}; reps=${4:-3}
    one() { (cd "$1" && timeout 300 python bench.py $flags --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('%-14s ms_per_step %.4f  value %.1f' % ('$1', d['ms_per_step'], d['value']))"); }
    cd "$R"
    for i in $(seq 1 $reps); do one "$tree"; one .; done
    ;;
*)  sed -n 2,10p "$0"; exit 2;;
esac
