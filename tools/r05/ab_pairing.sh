#!/bin/bash
# does the C2 loop gain from int32 pairing?  A two-operand int32 op costs a SIMD 2 cycles only when the NEXT
# wave the arbiter picks also has one (profiles/r03_valu_rates.txt: 4.5 cycles for one wave alone, 2.2 from two waves on); in the
# stepping loop's mix it measures 3.9.  Variants: the waves of a block re-aligned by s_barrier every trip / every 4th trip (two
# waves of a 512-thread block share each SIMD: in lock-step their Philox rounds would pair), and a scheduling fence after the
# Philox rounds (clustered runs).  Three alternating passes of tools/ubench/ab_kernels.py per variant on one box.
#   (build) bash tools/ubench/build_variants.sh tb1=-DSVMC_TRIP_BARRIER=1 tb4=-DSVMC_TRIP_BARRIER=4 fence=-DSVMC_PHILOX_FENCE=1
#   (run on the GPU box) bash tools/r05/ab_pairing.sh > gpurun_out/r05_ab_c2.jsonl
R=$(cd "$(dirname "$0")/../.." && pwd)
for pass in 1 2 3; do
  for v in cur tb1 tb4 fence; do
    timeout 120 python $R/tools/ubench/ab_kernels.py $R/tools/ubench/ab/libsvmc_$v.so $v 2>/dev/null | tail -1
  done
done
