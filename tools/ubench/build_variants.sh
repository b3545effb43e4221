#!/bin/bash
# builds the A/B libraries of tools/ubench/ab_kernels.py into tools/ubench/ab/ (git-ignored *.so; they travel to the GPU box)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/tools/ubench/ab
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -fno-gpu-rdc -mllvm --align-all-blocks=4 -DSVMC_BUILDING=1 -Wno-unused-function"
build() {  # name, source root, extra flags
  local name=$1 root=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS -I$root/include -I$root/stochvolmodels_amd/csrc "$@" $root/stochvolmodels_amd/csrc/svmc_*.hip -o $OUT/libsvmc_$name.so &
}
for b in ${BASES:-}; do                # previous commits as A sides: BASES="r1=f00a54f preqe=db02c7b"
  name=${b%%=*}; ref=${b#*=}
  rm -rf /tmp/svmc_$name && mkdir -p /tmp/svmc_$name && git -C $R archive $ref include stochvolmodels_amd/csrc | tar -x -C /tmp/svmc_$name
  build $name /tmp/svmc_$name
done
build cur $R
for v in "$@"; do                      # name=flags...   e.g. r10=-DSVMC_PHILOX_ROUNDS=10
  name=${v%%=*}; fl=${v#*=}
  build $name $R $fl
done
wait
ls -la $OUT
