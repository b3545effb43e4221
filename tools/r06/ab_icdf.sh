#!/bin/bash
# Round 6, VERDICT r05 item 2 ("one LDS read per normal"): A/B builds of the draw's table, timed on the C2 stepping kernel.
#   cur        the product: 1024 segments x cubic in |t| itself, fp64, 32 B per normal (two ds_read_b128), 8 VALU per normal
#   mixed      the same cubic in the EDGE form, {a0, a1} fp64 + {a2, a3} fp32: 24 B per normal (ds_read_b128 + ds_read_b64), 12 VALU
#   mixed_m6   ... with 2048 segments (48 KB of LDS per block)
#   m4         512 segments x cubic in |t|, fp64 (max error 1.1e-8): half the table, the same reads -- only for the vol-paths table
# Run here (build container): builds tools/ubench/ab/libsvmc_<name>.so; then on the GPU box: tools/r06/ab_icdf_run.sh
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/tools/ubench/ab
python $R/tools/gen_icdf_table.py --m 5 --deg 3 --edge --out $R/tools/ubench/ab/icdf_m5_edge.h
python $R/tools/gen_icdf_table.py --m 6 --deg 3 --edge --out $R/tools/ubench/ab/icdf_m6_edge.h
python $R/tools/gen_icdf_table.py --m 4 --deg 3 --raw --out $R/tools/ubench/ab/icdf_m4_raw.h
bash $R/tools/ubench/build_variants.sh \
  "mixed=-DSVMC_ICDF_MIXED=1 -DSVMC_ICDF_TABLE_HEADER=\"$R/tools/ubench/ab/icdf_m5_edge.h\"" \
  "mixed_m6=-DSVMC_ICDF_MIXED=1 -DSVMC_ICDF_TABLE_HEADER=\"$R/tools/ubench/ab/icdf_m6_edge.h\"" \
  "m4=-DSVMC_ICDF_TABLE_HEADER=\"$R/tools/ubench/ab/icdf_m4_raw.h\""
