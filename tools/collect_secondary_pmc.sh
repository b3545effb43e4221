#!/bin/bash
# SQ counters (own pass, kernel-trace only) for the secondary kernels: rough LogSV and Heston; run on the GPU box from
# the repo root, summarise with tools/rocpd_summary.py
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/prof_sq_rough -o rough -- python $R/tools/bench_rough.py > $R/gpurun_out/prof_sq_rough.log 2>&1; echo rough rc=$?
cat > /tmp/heston_once.py <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import stochvolmodels_amd as sv
kk = np.linspace(0.5, 1.5, 21); ty = np.where(kk >= 1, "C", "P"); ttms = np.array([0.25, 0.5, 0.75, 1.0])
for scheme in ("euler", "qe"):
    p = sv.HestonParams()
    for _ in range(2):
        sv.heston_mc_chain_pricer(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(kk,) * 4,
                                  optiontypes_ttms=(ty,) * 4, v0=p.v0, theta=p.theta, kappa=p.kappa, rho=p.rho,
                                  volvol=p.volvol, nb_path=1 << 22, scheme=scheme, nb_steps_per_year=508, seed=3)
PY
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/prof_sq_heston -o heston -- python /tmp/heston_once.py $R > $R/gpurun_out/prof_sq_heston.log 2>&1; echo heston rc=$?
