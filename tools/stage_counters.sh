#!/bin/bash
# SQ / LDS / memory-pipe counters for one stage of the path (separate rocprofv3 --pmc passes of tools/kbench.py,
# kernel-trace only beside the counters):  tools/stage_counters.sh <tag> <stage> [nq]
#   -> gpurun_out/<tag>_<stage>_<pass>/...csv ; condense with
#   tools/summarise_counters.py <tag> <kernel-substring> <stage>   -> profiles/<tag>_<stage>_counters.csv
set -e
TAG=${1:-r02}
STAGE=${2:-bucket}
NQ=${3:-10000000}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$O/${TAG}_${STAGE}_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o run -- python3 $R/tools/kbench.py --stage $STAGE --nq $NQ --reps 2 > $D.log 2>&1 || echo "pass $i ($C) failed"
  find $D -type f ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' -delete 2>/dev/null || true
  echo "pass $i: $C"
done
