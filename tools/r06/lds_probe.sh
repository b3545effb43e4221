#!/bin/bash
# Round 6: what the table reads cost the generators.  Three measurement builds of the library (SVMC_PROBE bit 0: the step's exp-table
# read replaced by the constant 1.0; bit 1: the draw's two ds_read_b128 per normal replaced by constants; 3: both -- the results are
# garbage, the instruction streams are the product's minus those LDS instructions) against the product, wall time of
# logsv_mc_chain_pricer through tools/r06/mid_waves_sweep.py: C2's launch on the full-launch kernels, and the few-waves kernels.
#   bash tools/r06/lds_probe.sh          (on a GPU box; builds into build/, writes gpurun_out/r06_probe_*.json)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build gpurun_out
SRC="stochvolmodels_amd/csrc"
for n in 1 2 3; do
  if [ ! -f build/probe_$n.so ]; then
    (mkdir -p build/p$n && cd build/p$n && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
       -fno-gpu-rdc -mllvm --align-all-blocks=4 -I../../include -I../../$SRC -DSVMC_BUILDING=1 -DSVMC_PROBE=$n -Wno-unused-function \
       -Wno-pass-failed ../../$SRC/svmc_runtime.hip ../../$SRC/svmc_kernels.hip ../../$SRC/svmc_analytic.hip ../../$SRC/svmc_chain.hip \
       ../../$SRC/svmc_comm.hip ../../$SRC/svmc_multi.hip -o ../probe_$n.so) &
  fi
done
wait
for rep in a b; do
  for n in 0 1 2 3; do
    if [ $n = 0 ]; then L=$PWD/stochvolmodels_amd/libsvmc.so; else L=$PWD/build/probe_$n.so; fi
    SVMC_LIB=$L SVMC_ALLOW_OLD_ABI=1 python tools/r06/mid_waves_sweep.py 80 --variants -1 --cases c2 > gpurun_out/r06_probe_c2_${n}${rep}.json
  done
done
for n in 0 1 2 3; do
  if [ $n = 0 ]; then L=$PWD/stochvolmodels_amd/libsvmc.so; else L=$PWD/build/probe_$n.so; fi
  SVMC_LIB=$L SVMC_ALLOW_OLD_ABI=1 python tools/r06/mid_waves_sweep.py 60 --variants 0 --cases one > gpurun_out/r06_probe_${n}.json
done
