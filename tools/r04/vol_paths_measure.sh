#!/bin/bash
# On the GPU box (via gpurun), from the repo root: the measurements behind profiles/r04_vol_paths.json.
#   tools/ubench/build_variants.sh must have built r3 (round-3 library), cur, blk512, probe1 (no stores), probe2 (stores to one
#   row) into tools/ubench/ab/, and tools/r04/write_bw must be compiled.
set -u
R=$PWD
O=$R/gpurun_out/r04_vol_paths
rm -rf $O; mkdir -p $O
tools/r04/write_bw > $O/write_bw.jsonl 2> $O/write_bw.err
for rep in 1 2; do
  for v in r3 cur blk512 probe1 probe2; do
    python tools/ubench/ab_vol_paths.py tools/ubench/ab/libsvmc_$v.so $v >> $O/ab.jsonl 2>> $O/ab.err
  done
done
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o vp -- python $R/tools/ubench/ab_vol_paths.py $R/tools/ubench/ab/libsvmc_cur.so cur > $O/write.log 2>&1; echo write rc=$?
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o vp -- python $R/tools/ubench/ab_vol_paths.py $R/tools/ubench/ab/libsvmc_cur.so cur > $O/fetch.log 2>&1; echo fetch rc=$?
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o vp -- python $R/tools/ubench/ab_vol_paths.py $R/tools/ubench/ab/libsvmc_cur.so cur > $O/kt.log 2>&1; echo kt rc=$?
find $O -type f ! -name '*.db' ! -name '*.log' ! -name '*.jsonl' ! -name '*.err' ! -name '*.csv' -delete
cd $R
