#!/bin/bash
# Root-causing the one-time ~60 ms stall: trace the HIP and HSA API calls of a C2 process whose
# timed region contains the slow chain call, list the API calls that take more than 5 ms with their neighbours.
# Run on the GPU box from the repo root (via gpurun); output: gpurun_out/stall_{hip,hsa}.txt
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
for DOM in hip hsa; do
  D=/tmp/stall_$DOM
  rm -rf $D
  SVMC_BENCH_PREWARM=0 timeout 600 rocprofv3 --$DOM-trace --kernel-trace --output-format csv -d $D -o t -- \
      python $R/bench.py --config c2 --steps 220 --warmup 0 --no-cpu-baseline --no-extra-legs --no-streamed > $D.log 2>&1
  echo "$DOM rc=$?"
  tail -2 $D.log | cut -c1-1500 > $R/gpurun_out/stall_$DOM.txt
  python $R/tools/r03/stall_scan.py $D >> $R/gpurun_out/stall_$DOM.txt 2>&1
done
cd $R
