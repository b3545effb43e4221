# (goes with commit e53a51a, where the dynamic-units launch lived: SVMC_UNIT_* are read by that build only)
# A/B of the dynamic-units launch against the one-round kernel (tools/ubench/build_variants.sh builds the libraries)
run() { # lib, env...
  local lib=$1; shift
  echo "== $lib $*"
  env SVMC_PROBE_LIB=tools/ubench/ab/libsvmc_$lib.so "$@" timeout 60 python tools/r03/units_probe.py 20 21 2>&1 | grep -v amdgpu.ids | cut -c1-900
}
SVMC_UNITS_DEBUG=1 SVMC_UNIT_SPIN_LIMIT=200000 run cur SVMC_UNIT_STEPS=256 | tail -3
for rep in 1 2; do
  run cur SVMC_UNIT_STEPS=0
  run cur SVMC_UNIT_STEPS=256
  run cur SVMC_UNIT_STEPS=256 SVMC_UNIT_QUEUES=1
  run cur SVMC_UNIT_STEPS=128
  run cur SVMC_UNIT_STEPS=512
  run cur SVMC_UNIT_STEPS=1024 SVMC_UNIT_FORCE=1
  run pf SVMC_UNIT_STEPS=256
  run noprio SVMC_UNIT_STEPS=256
  run sg96 SVMC_UNIT_STEPS=256
done
