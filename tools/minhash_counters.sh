#!/bin/bash
# Counter evidence for the MinHash kernel: separate rocprofv3 --pmc passes of
# tools/kbench.py --stage minhash (kernel-trace only beside the counters), one directory per pass.
#   tools/minhash_counters.sh <tag> [nq]      -> gpurun_out/<tag>_mh_<pass>/...csv
# tools/summarise_counters.py <tag> condenses them into profiles/<tag>_minhash_counters.csv.
set -e
TAG=${1:-r02}
NQ=${2:-1000000}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
i=0
for C in "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" \
         "GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  D=$O/${TAG}_mh_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o run -- python3 $R/tools/kbench.py --stage minhash --nq $NQ --reps 3 > $D.log 2>&1 || echo "pass $i ($C) failed"
  find $D -type f ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' -delete 2>/dev/null || true
  echo "pass $i: $C"
done
