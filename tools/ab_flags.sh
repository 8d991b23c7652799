#!/bin/bash
# A/B of compile-time switches on ONE box: tools/ab_flags.sh "<flags A>" "<flags B>" ...  (development tool)
# rebuilds libqrlsh.so with each set of -D flags and runs tools/kbench.py at 10 M and 1 M queries (or $AB_CMD).
set -e
R=/root/repo
cd $R/query-recommendation-system_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function"
i=0
for F in "$@"; do
  i=$((i+1))
  touch *.hip
  make -j16 CXXFLAGS="$BASE $F" > /dev/null 2>&1
  if [ -n "$AB_CMD" ]; then
    (cd $R && $AB_CMD > $R/gpurun_out/ab_${i}_cmd.log 2>&1)
  else
    python3 $R/tools/kbench.py --nq 10000000 --reps 6 > $R/gpurun_out/ab_${i}_10M.log 2>&1
    python3 $R/tools/kbench.py --nq 1000000 --reps 12 > $R/gpurun_out/ab_${i}_1M.log 2>&1
  fi
  echo "variant $i: $F"
done
