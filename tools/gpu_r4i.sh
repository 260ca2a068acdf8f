#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python tools/bf16_drift.py > $O/bf16_drift.log 2>&1; grep -v amdgpu.ids $O/bf16_drift.log | tail -60
for m in alone after_writer after_other after_idle; do timeout 200 python tools/index_max_instep.py $m 2>&1 | grep index_max_gather; done | tee $O/index_max_instep.log
cd /tmp
for m in alone after_writer after_other; do
  for cset in "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    t=${m}_$(echo $cset | cut -c1-8 | tr ' ' '_')
    rocprofv3 --pmc $cset --kernel-trace --output-format csv -d $O/pmc_$t -o pmc -- python $R/tools/index_max_instep.py $m 5 > /dev/null 2> $O/pmc_$t.err
  done
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; du -sh $O
