#!/bin/bash
# counters of the generator kernels for a list of library variants (tools/ubench/ab/libsvmc_<v>.so): one rocprofv3 --pmc
# pass per variant over tools/ubench/ab_kernels.py, summarised by tools/rocpd_summary.py into gpurun_out/r03_ab_pmc_<v>.txt
set -u
R=$PWD
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  D=/tmp/abpmc_$v; rm -rf $D
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
      --kernel-trace -d $D -o t -- python $R/tools/ubench/ab_kernels.py $R/tools/ubench/ab/libsvmc_$v.so $v > $D.log 2>&1
  echo "$v rc=$?"
  python $R/tools/rocpd_summary.py $(find $D -name '*.db') > $R/gpurun_out/r03_ab_pmc_$v.txt 2>&1
done
