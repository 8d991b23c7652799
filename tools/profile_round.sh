#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (configs[2]: 10 M queries, one GPU)
# 2. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (never combined with other trace domains)
# 3. a plain bench.py run (with the CPU baseline leg) -> gpurun_out/<tag>_bench.json
# tools/summarise_profile.py <tag> ... then condenses them into profiles/ (run in the repo afterwards).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o run -- python3 $R/bench.py --cpu-sample 0 --no-secondary > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace.err
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -o run -- python3 $R/bench.py --cpu-sample 0 --no-secondary --steps 3 --warmup 1 --no-prof > $O/${TAG}_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -o run -- python3 $R/bench.py --cpu-sample 0 --no-secondary --steps 3 --warmup 1 --no-prof > $O/${TAG}_pmc_write.json 2> $O/${TAG}_pmc_write.err
echo "write done"
# keep only the small CSVs (the merged gpurun_out/ is capped at 64 MiB)
find $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -type f ! -name '*.csv' -delete 2>/dev/null || true
find $O/${TAG}_trace -name '*kernel_trace.csv' -delete 2>/dev/null || true
cd $R && python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cat $O/${TAG}_bench.json | head -c 1500
