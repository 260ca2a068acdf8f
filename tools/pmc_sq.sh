#!/bin/bash
# tools/pmc_sq.sh <one_kernel name> -- SQ counters of one hot kernel (two --pmc passes, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-fused_pool}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_${K}_a -o pmc -- python $R/tools/one_kernel.py $K 3 > /dev/null 2> $R/gpurun_out/pmc_sq_${K}_a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_MISC \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_${K}_b -o pmc -- python $R/tools/one_kernel.py $K 3 > /dev/null 2> $R/gpurun_out/pmc_sq_${K}_b.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_FLAT \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_${K}_c -o pmc -- python $R/tools/one_kernel.py $K 3 > /dev/null 2> $R/gpurun_out/pmc_sq_${K}_c.err
for p in a b c; do tail -3 $R/gpurun_out/pmc_sq_${K}_$p.err; done
