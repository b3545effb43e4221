#!/bin/bash
# Round 6, VERDICT r05 weak item 4a: the DRAWING vol-paths kernel (2^20 paths x 1024 steps, 8.6 GB written) in the variants round 5
# listed and did not run.  Build here (build container):  bash tools/r06/ab_vol_paths.sh build
# Run on the GPU box:                                       bash tools/r06/ab_vol_paths.sh run   -> gpurun_out/r06_vol_paths_ab.jsonl
#   cur      the product
#   ahead    the next Philox call's eight table reads in flight under this call's four steps (same bits)
#   burst    four steps into registers, then four stores back to back (same bits)
#   b512     512-thread blocks (four per CU instead of two 1024-thread ones)
#   m4       a 512-segment table (16 KB, max error 1.1e-8: other bits) -- "cheaper draw"
#   nostore  the arithmetic alone (the stores behind a test that never holds)
#   l2store  every store on one row (L2-resident): arithmetic + store issue without HBM
# per variant: ms, written TB/s, the shader clock measured inside the launch, and the socket power / clock rocm-smi shows while
# the kernel repeats for four seconds (tools/r04/power_probe.py, its "vol paths, device RNG" case)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
AB=$R/tools/ubench/ab
if [ "${1:-}" = build ]; then
  mkdir -p $AB
  python $R/tools/gen_icdf_table.py --m 4 --deg 3 --raw --out $AB/icdf_m4_raw.h
  bash $R/tools/ubench/build_variants.sh "ahead=-DSVMC_VOLPATHS_VARIANT=1" "burst=-DSVMC_VOLPATHS_VARIANT=2" "b512=-DSVMC_VOLPATHS_RNG_BLOCK=512" \
      "m4=-DSVMC_ICDF_TABLE_HEADER=\"$AB/icdf_m4_raw.h\"" "nostore=-DSVMC_VOLPATHS_PROBE=1" "l2store=-DSVMC_VOLPATHS_PROBE=2"
  exit 0
fi
cd $R
mkdir -p gpurun_out
: > gpurun_out/r06_vol_paths_ab.jsonl
for v in cur ahead burst b512 m4 nostore l2store; do
  a=$(timeout 300 python tools/ubench/ab_vol_paths.py $AB/libsvmc_$v.so $v 2>/dev/null | tail -1)
  b=$(SVMC_LIB=$AB/libsvmc_$v.so timeout 120 python tools/r04/power_probe.py 2>/dev/null | grep 'vol paths, device RNG')
  echo "{\"variant\": \"$v\", \"timing\": $a, \"power\": ${b:-null}}" >> gpurun_out/r06_vol_paths_ab.jsonl
done
