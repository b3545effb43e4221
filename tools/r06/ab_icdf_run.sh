#!/bin/bash
# on the GPU box, after tools/r06/ab_icdf.sh built the variants: three alternating passes of tools/ubench/ab_kernels.py per variant
# (kernel times by HIP events) -> gpurun_out/r06_ab_icdf.jsonl, then one counter pass per variant (LDS / VALU) ->
# gpurun_out/r06_ab_icdf_pmc_<v>.txt
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
VARIANTS="${@:-cur mixed mixed_m6}"
: > $R/gpurun_out/r06_ab_icdf.jsonl
for pass in 1 2 3; do
  for v in $VARIANTS; do
    timeout 300 python tools/ubench/ab_kernels.py tools/ubench/ab/libsvmc_$v.so $v 2>/dev/null | tail -1 >> $R/gpurun_out/r06_ab_icdf.jsonl
  done
done
cd /tmp
for v in $VARIANTS; do
  D=/tmp/abpmc_$v; rm -rf $D
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
      --kernel-trace -d $D -o t -- python $R/tools/ubench/ab_kernels.py $R/tools/ubench/ab/libsvmc_$v.so $v > $D.log 2>&1
  echo "$v pmc rc=$?"
  python $R/tools/rocpd_summary.py $(find $D -name '*.db') 2>&1 | grep -E "logsv_rng_kernel|^kernel" | cut -c1-150 > $R/gpurun_out/r06_ab_icdf_pmc_$v.txt
done
