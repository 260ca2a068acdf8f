#!/bin/bash
# round 4, second GPU call: parity of the reworked third-generation layer, the sweep, counters of the 320 -> 384 layer
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_h3p.py -x -q > $O/pytest_h3p.log 2>&1; tail -4 $O/pytest_h3p.log
V=$R/so-net_amd/lib/libsonet_hip_variants.so
SONET_HIP_LIB=$V timeout 900 python tools/bench_h3p.py --sweep --shapes big > $O/bench_h3p_sweep_big.log 2>&1; tail -3 $O/bench_h3p_sweep_big.log
SONET_HIP_LIB=$V timeout 300 python tools/bench_h3p.py --shapes all > $O/bench_h3p_default.log 2>&1
cd /tmp
pmc() {  # tag, counters..., then env for the run
  tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o pmc -- python $R/tools/one_h3p.py $NAME 3 > /dev/null 2> $O/pmc_$tag.err
}
export SONET_HIP_LIB=$V
for NAME in 320x384 1024x512; do
  export NAME
  for cfg in "4,2,2:0" "12,1,1:1"; do
    shp=${cfg%%:*}; ns=${cfg##*:}
    [ "$NAME" = "1024x512" ] && [ "$shp" = "12,1,1" ] && continue
    export SONET_H3P_SHAPE=$shp
    if [ "$ns" != "0" ]; then export SONET_H3P_NSLAB=$ns; else unset SONET_H3P_NSLAB; fi
    t=${NAME}_${shp//,/_}
    pmc ${t}_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
    pmc ${t}_b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE
    pmc ${t}_f FETCH_SIZE
    pmc ${t}_w WRITE_SIZE
    pmc ${t}_c TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  done
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; du -sh $O
