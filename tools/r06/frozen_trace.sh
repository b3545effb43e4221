#!/bin/bash
# kernel trace of one calibration objective evaluation on frozen randoms (tools/r05/bench_frozen.py, 10^5 paths x 364 steps), with
# the round-6 two-launch tail and with round 5's five-node tail (SVMC_CHAIN_TAIL_NODES=5)
export TMPDIR=/tmp SVMC_BENCH_FROZEN_ONLY=1
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp
for t in 2 5; do
  rm -rf /tmp/ft_$t
  SVMC_CHAIN_TAIL_NODES=$t timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ft_$t -o fr -- python $R/tools/r05/bench_frozen.py 100000 100 > /tmp/ft_$t.log 2>&1
  echo "## SVMC_CHAIN_TAIL_NODES=$t"; python $R/tools/rocpd_summary.py $(find /tmp/ft_$t -name '*.db' | head -1)
done > $R/gpurun_out/r06_frozen_trace.txt
