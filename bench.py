#!/usr/bin/env python
"""
bench.py -- MC path-steps/s of the StochVolModels Monte Carlo hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4]
        N > 1 without a launcher: bench.py starts its own N ranks (python -m torch.distributed.run --nnodes=1
        --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>), one per visible GPU, RCCL backend
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (the driver's form: used as launched)

Workloads (BASELINE.json `configs`):
  c2  (default at --gpus 1; the configuration the headline metric is quoted on)  LogSV quadratic-drift MC,
      LOGSV_BTC_PARAMS, 2^20 paths per GPU x 1024 log-Euler steps (ttm = 1, 1023 steps/yr -> int(1023)+1 = 1024,
      dt = 2^-10), one expiry x 21 strikes (linspace(0.5, 1.5, 21), puts below the forward, calls at/above).
  c4  (default at --gpus N > 1)  LogSV, LOGSV_BTC_PARAMS, 2^21 paths per GPU (2^24 = "16M" at 8 GPUs), 8 expiries
      ttm = k/8 at 1016 steps/yr -> 8 x 128 = 1024 steps, forwards 67000 e^{0.05 T}, discount factors e^{-0.05 T},
      21 strikes per expiry at F linspace(0.6, 1.6, 21), puts below the forward, calls at/above; paths sharded by
      global path id, the two packed all-reduces over RCCL per chain.
Both: spot measure, LOG_RETURN, on-device counter-based randoms.  One "step" = one complete
logsv_mc_chain_pricer call: state init, stepping kernel(s), spot-sum and payoff reductions, (N>1: the two
all-reduces), D2H of the prices.  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line (rank 0): `value` = whole-job path-steps/s; `roofline` = the dominant kernel against the roof
that binds it (VALU issue: the kernel moves 56 B per path per expiry and no HBM byte inside the time loop): the time
loop's instruction histogram read off the compiler's assembly of the loaded library (libsvmc.isa.json, written by the
build) priced per opcode class with the issue costs measured in profiles/r03_valu_rates.txt, over the kernel time from
HIP events on the launch stream; `roofline_hbm` = the same kernel against HBM (BASELINE asks for it); `cpu_baseline` =
the CPU oracle (a port of the reference's algorithm) timed on one host core on a bounded sample; at N > 1 also
`n1_share_value` (this rank's shard priced WITHOUT the group: the denominator of the weak-scaling ratio),
`c4_full_one_gpu` (all N x 2^21 paths on ONE device: the denominator of the strong-scaling ratio), `rccl_ranks_seen` and
`rccl_route` (the same chain timed through libsvmc's own RCCL entry points, the route a C host takes).
Every line carries `c_abi_route`: the same workload through ONE call of the fused C driver svmc_logsv_chain_price per step
(the boundary's headline entry point; include/svmc.h), its prices compared with the Python route's.

The collective layer of an N > 1 run is chosen by a LADDER (stochvolmodels_amd.dist.init_with_fallback): a gloo control
plane first, then torch's RCCL backend -> libsvmc's own RCCL entry points -> the gloo group itself, each rung probed in a child
process under a deadline before the ranks commit to it -- a node whose RCCL fails or hangs still yields a number, flagged
(`comm_ladder.rung`, `comm_fallback_reason`).  `single_process_route` = the same job by ONE process driving all N devices
(svmc_multi_*: a host thread and a session per device), run as a child of rank 0 after the timed region.
`python bench.py --gpus N --single-process` times that route on its own.
At N = 1 the line ends with `secondary`: the other BASELINE configurations in compact form -- C1, C3 (Euler and QE, both
parameter sets), C5 (analytic batch, Monte Carlo per set, the reference's 4-stderr verdict), the frozen-randoms calibration
objective (f3) and C2's rate at 2^21 paths -- each with a parity scalar against the oracle on the same stream at a reduced size.

The N > 1 line PROVES ITSELF (exit status != 0 when a proof fails, the line still printed with `self_check.failed`):
`sharded_vs_one_gpu_max_rel_dev` -- the sharded job and the whole job on rank 0's device, same seed, must agree to
reduction-order rounding (<= 1e-12); on the RCCL backend `rccl_ranks_seen` must equal --gpus; `kernel_ms_over_ranks`
shows a slow GPU.  A rank that never arrives at the rendezvous, an RCCL initialisation that does not return within
SVMC_BENCH_INIT_TIMEOUT (120 s) or a leg that overruns its deadline ends the rank with its traceback on stderr (and
torch.distributed.run then ends the others): the command returns non-zero instead of hanging.
See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import faulthandler
import gc
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before HIP initialises

import numpy as np  # noqa: E402

# MI355X (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8 TB/s spec; 256 CUs x 4 SIMDs, max clock 2400 MHz; vector fp64
# 78.6 TFLOP/s (FMA = 2), i.e. one fp64 wave64 instruction per SIMD per 4 cycles
HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TFLOPS = 78.6
N_SIMD = 256 * 4
MAX_CLOCK_HZ = 2.4e9
SIMD_CYCLES_PEAK = N_SIMD * MAX_CLOCK_HZ               # SIMD issue cycles per second, whole chip
FP64_VECTOR_PEAK_FLOPS = 78.6e12                       # MI355X_MICROARCH.md: fp64 vector (non-MFMA) peak, FMA = 2 flop
# Issue cost of a wave64 VALU instruction on one SIMD, in shader cycles, by opcode class -- MEASURED on this chip with
# tools/ubench/valu_rates.hip (profiles/r03_valu_rates.txt, s_memtime ticks, 8 waves per SIMD; measured 4.08-4.17 / 16.2 /
# 4.07-4.17 / 2.17-2.25 / 8.1, rounded DOWN to the architectural figure so that the roof is never understated):
#   fp64       4   v_fma / mul / add / max / ldexp / cvt ..._f64, v_mad_u64_u32, 64-bit shifts and moves
#   quarter   16   v_rcp_f64, v_rsq_f64, v_sqrt_f64
#   int32_3op  4   the three-operand 32-bit integer ops (v_bfi, v_add3, v_lshl_add, v_perm, v_alignbit, v_mul_lo/hi ...)
#   int32      2   two-operand 32-bit ops, v_bitop3_b32, v_fma_f32 -- 2 only back to back; mixed into an fp64 stream they
#                  cost 3.5 (1:1) to 3.9 (1:3) cycles each (the MIX rows of the same file): `frac` below prices them at 2,
#                  `frac_in_stream_int32_cost` at the 3.9 the stepping loop's own mix (about 1 in 4) measures
#   trans32    8   v_exp / log / rcp / rsq / sqrt / sin / cos _f32
CLASS_CYCLES = {"fp64": 4.0, "quarter": 16.0, "int32_3op": 4.0, "int32": 2.0, "trans32": 8.0}
INT32_IN_FP64_STREAM_CYCLES = 3.9
# Untimed steps ahead of the caller's warm-up (disclosed as device_prewarm_steps): the GPU's clock ramp -- the first ~10
# launches of a process run 20-25 % slow.  (Round 2 ran 200 here to get past a one-time ~60 ms step; that step is CPython's
# generation-2 garbage collection -- profiles/r03_gc_stall.json: gc callbacks put a 53-67 ms full collection exactly there
# -- and the timed region now runs after gc.collect() + gc.freeze(), as any latency-sensitive Python service does.)
PREWARM = int(os.environ.get("SVMC_BENCH_PREWARM", "10"))         # the environment override exists for the test suite
# (SURVEY.md 8d's flop-equivalent ESTIMATE -- 140 op-eq per path-step, a model of the reference's Box-Muller arithmetic -- is no
# longer printed: the kernel draws by table inversion and the model's "fraction" exceeded 1 by construction)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=("c2", "c4"), default=None, help="default: c2 at --gpus 1, c4 at --gpus N > 1")
    ap.add_argument("--paths-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip n1_share / c4_full_one_gpu / all-cores legs")
    ap.add_argument("--no-c-abi-route", action="store_true", help="skip the fused C driver leg (svmc_logsv_chain_price)")
    ap.add_argument("--no-self-check", action="store_true",
                    help="N > 1: skip the sharded-vs-one-GPU price comparison (it prices the whole job on rank 0's device)")
    ap.add_argument("--cpu-sample-paths", type=int, default=1 << 19)
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all --gpus devices (svmc_multi_*: a host thread and a session per device)")
    ap.add_argument("--reduce", choices=("auto", "host", "rccl"), default="auto",
                    help="--single-process: transport of the two all-reduces (auto = RCCL when its probe passes, else host)")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device ids (default 0..N-1)")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the `secondary` block (C1, C3, C5, f3)")
    ap.add_argument("--no-single-process-leg", action="store_true", help="N > 1: skip the single_process_route leg")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def make_workload(name: str, sv):
    """-> dict(chain arrays + spy) of a BASELINE configuration"""
    if name == "c2":
        ttms = np.array([1.0])
        forwards, dfs = np.ones(1), np.ones(1)
        kk = np.linspace(0.5, 1.5, 21)
        strikes = (kk,)
        types = (np.where(kk >= 1.0, "C", "P"),)
        spy = 1023
        label = "C2 LogSV quadratic-drift MC, LOGSV_BTC_PARAMS, on-device counter-based randoms"
    else:
        ttms = np.arange(1, 9) / 8.0
        forwards = 67000.0 * np.exp(0.05 * ttms)
        dfs = np.exp(-0.05 * ttms)
        strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in forwards)
        types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, forwards))
        spy = 1016
        label = "C4 LogSV MC, LOGSV_BTC_PARAMS, 8-expiry chain ttm = k/8, path-sharded, on-device counter-based randoms"
    grids, t0 = [], 0.0
    for t in ttms:
        nb, dt, _ = sv.set_time_grid(t - t0, spy)
        grids.append((nb, dt))
        t0 = t
    return dict(name=name, label=label, ttms=ttms, forwards=forwards, dfs=dfs, strikes=strikes, types=types, spy=spy,
                grids=grids, nb_total=sum(g[0] for g in grids), n_strikes=sum(len(k) for k in strikes))


def price(sv, wl, P, n_path, seed, comm=None, devices=None, reduce=None):
    return sv.logsv_mc_chain_pricer(ttms=wl["ttms"], forwards=wl["forwards"], discfactors=wl["dfs"],
                                    strikes_ttms=wl["strikes"], optiontypes_ttms=wl["types"], v0=P.sigma0, theta=P.theta,
                                    kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta, volvol=P.volvol,
                                    vol_backbone_etas=np.ones(len(wl["ttms"])), nb_path=n_path,
                                    nb_steps_per_year=wl["spy"], seed=seed, comm=comm, devices=devices, reduce=reduce)


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle = the checker of tests/, here only timed)
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, n_sample: int, P) -> dict:
    """the reference's algorithm on ONE host core (the reference is single-threaded): per expiry materialise W0, W1
    [nb_steps_i, n] with MT19937 + polar normals (NumPy's legacy RandomState == the reference's generator family,
    get_randoms_for_chain_valuation order), then the step-major fp64 loop with the state carried slice to slice and
    the payoff pass -- oracle/svmc_oracle.c (kind "port").  Done in chunks of 2^17 paths (2.1 GB of normals at 1024
    steps), like BASELINE.md section 3 prescribes."""
    from oracle import oracle
    oracle.build()
    chunk = min(n_sample, 1 << 17)
    t_rng = t_all = 0.0
    rng = np.random.RandomState(10)
    done = 0
    pooled = []
    while done < n_sample:
        x, s, q = np.zeros(chunk), P.sigma0 * np.ones(chunk), np.zeros(chunk)
        first = None
        for i, (nb, dt) in enumerate(wl["grids"]):
            t0 = time.perf_counter()
            W0 = rng.normal(0, 1, size=(nb, chunk))
            W1 = rng.normal(0, 1, size=(nb, chunk))
            t1 = time.perf_counter()
            x, s, q = oracle.logsv_terminal_w(x, s, q, dt, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, W0, W1)
            pr, sd = oracle.payoff(x, q, float(wl["ttms"][i]), float(wl["forwards"][i]), wl["strikes"][i], wl["types"][i],
                                   float(wl["dfs"][i]))
            t2 = time.perf_counter()
            first = pr if first is None else first
            t_rng += t1 - t0
            t_all += t2 - t0
            del W0, W1
        pooled.append(first)
        done += chunk
    return {"value": done * wl["nb_total"] / t_all, "unit": "path-steps/s", "cores": 1, "kind": "port",
            "sample": f"{done} paths x {wl['nb_total']} steps ({len(wl['grids'])} expiries) in chunks of {chunk}, "
                      f"{wl['n_strikes']} strikes; RandomState normals {t_rng:.1f}s of {t_all:.1f}s; "
                      f"host cores available: {os.cpu_count()}",
            # mean over the chunks; NB the estimator recentres by the SAMPLE mean of S_T, which is heavy-tailed under the
            # BTC parameters, so chunk prices scatter by more than the reported stderr (parity with the GPU is
            # established on identical randoms in tests/test_gpu_fullsize.py, not from these independent samples)
            "prices_head": [float(v) for v in np.mean(pooled, axis=0)[:3]],
            "prices_head_chunk_scatter": [float(v) for v in np.std(pooled, axis=0)[:3]]}


def cpu_baseline_numpy(nb_steps: int, P, n: int = 1 << 15) -> dict:
    """SURVEY.md 8d's other CPU number: the NumPy restatement of the reference's array code (oracle.np_logsv_terminal_w:
    the step-major loop of pricers/logsv_pricer.py:1040-1045 on whole [nb_path] vectors with NumPy temporaries, i.e.
    what the reference executes when Numba is absent), one core, RandomState normals, a small bounded sample"""
    from oracle import oracle
    rng = np.random.RandomState(11)
    t0 = time.perf_counter()
    W0 = rng.normal(0, 1, size=(nb_steps, n))
    W1 = rng.normal(0, 1, size=(nb_steps, n))
    oracle.np_logsv_terminal_w(np.zeros(n), P.sigma0 * np.ones(n), np.zeros(n), 1.0 / nb_steps, P.theta, P.kappa1, P.kappa2,
                               P.beta, P.volvol, W0, W1)
    t = time.perf_counter() - t0
    try:
        import numba  # noqa: F401
        have_numba = True
    except Exception:
        have_numba = False
    return {"value": n * nb_steps / t, "unit": "path-steps/s", "cores": 1, "kind": "port (NumPy array code)",
            "sample": f"{n} paths x {nb_steps} steps incl. RandomState normals, {t:.1f}s",
            "numba_importable_on_this_box": have_numba}


def cpu_baseline_all_cores(nb_steps: int, P) -> dict:
    """the same fp64 step on ALL host cores with the counter-based draw generated on the fly, OpenMP over paths
    (oracle svo_logsv_terminal_rng) -- not the reference's algorithm (which is serial), but the fairest CPU number
    for the GPU kernel's own algorithm."""
    from oracle import oracle
    oracle.build()
    cores = oracle.set_threads(oracle.effective_cores())      # affinity capped by the cgroup quota
    n = min(1 << 22, max(1 << 14, (cores * (1 << 16))))
    x0, s0, q0 = np.zeros(n), P.sigma0 * np.ones(n), np.zeros(n)
    t0 = time.perf_counter()
    oracle.logsv_terminal_rng(x0, s0, q0, nb_steps, 1.0 / nb_steps, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 7)
    t = time.perf_counter() - t0
    return {"value": n * nb_steps / t, "unit": "path-steps/s", "cores": cores, "kind": "port (counter-based draw, OpenMP)",
            "sample": f"{n} paths x {nb_steps} steps, stepping only, {t:.1f}s; os.cpu_count() = {os.cpu_count()}"}


# ---------------------------------------------------------------------------------------------------------------
# profile-derived constants (committed under profiles/, collected by tools/collect_profiles.sh)
# ---------------------------------------------------------------------------------------------------------------
def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def load_isa(lib_path: str) -> dict:
    """the time loops' instruction histograms of the LOADED library: libsvmc.isa.json is written beside libsvmc.so by the
    build that produced it (stochvolmodels_amd/build.py, from that compilation's own assembly) and names the library's
    sha256 -- a mismatch (someone rebuilt the .so by hand) marks every roofline of the line stale"""
    out = {"lib_sha256": file_sha256(lib_path), "kernels": {}, "stale": True, "source": None}
    path = os.path.join(os.path.dirname(lib_path), "libsvmc.isa.json")
    try:
        with open(path) as fh:
            isa = json.load(fh)
    except (OSError, ValueError):
        return out
    out["kernels"] = isa.get("kernels", {})
    out["stale"] = isa.get("lib_sha256") != out["lib_sha256"]
    out["source"] = os.path.relpath(path, ROOT)
    return out


def load_pmc(lib_sha256: str) -> dict:
    """the committed counter passes (rocprofv3 --pmc; tools/collect_profiles.sh -> profiles/rNN_pmc.json): measured
    SQ_INSTS_VALU per wave-step (cross-check of the assembly count), the LDS / VALU busy counters and HBM traffic.  Each
    file was collected on ONE build and names its sha256: the file of the loaded library is taken when there is one
    (`matches_loaded_library`), else the newest round's -- then only its traffic figures are quoted (they do not depend on
    the instruction stream), never its instruction counters"""
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")), reverse=True):
        try:
            with open(path) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        pmc["file"] = os.path.relpath(path, ROOT)
        pmc["matches_loaded_library"] = pmc.get("lib_sha256") == lib_sha256
        if pmc["matches_loaded_library"]:
            return pmc
        best = best or pmc
    return best


class ClockPoller:
    """best-effort engine clock of this rank's GPU while the timed region runs (amdgpu sysfs pp_dpm_sclk: the line
    marked '*'); None when the node does not expose it"""

    def __init__(self, device_index: int):
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.path = cands[device_index] if device_index < len(cands) else (cands[0] if cands else None)
        self.samples, self._stop, self._t = [], threading.Event(), None

    def _read(self):
        try:
            for line in open(self.path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError, TypeError):
            return None
        return None

    def __enter__(self):
        if self.path is not None:
            def run():
                while not self._stop.wait(0.1):
                    v = self._read()
                    if v:
                        self.samples.append(v)
            self._t = threading.Thread(target=run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t is not None:
            self._t.join()

    def steady_mhz(self):
        """median of the busy readings in the second half of the samples (the first half covers the sensor's lag; a
        reading below 500 MHz is the idle state the sensor reports between refreshes); None if there is none"""
        half = [v for v in self.samples[len(self.samples) // 2:] if v > 500.0]
        return float(np.median(half)) if half else None


def streamed_roofline(eng, P, nb_steps: int, pmc) -> dict:
    """the fixed-randoms kernel (logsv_w_kernel) reads 16 B per path-step from HBM: the HBM-bound leg."""
    n = eng.n_path
    w0, w1 = eng.fill_normals(nb_steps, 99)
    eng.fill_state(0.0, P.sigma0, 0.0)
    dt = 1.0 / 1024
    for _ in range(2):
        eng.logsv_w(nb_steps, dt, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1.0, True, w0, w1)
    eng.synchronize()
    eng.start_kernel_timing()
    for _ in range(5):
        eng.fill_state(0.0, P.sigma0, 0.0)
        eng.logsv_w(nb_steps, dt, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1.0, True, w0, w1)
    ms = float(np.mean(eng.stop_kernel_timing()["logsv_w_kernel"]))
    alg_bytes = (16.0 * nb_steps + 48.0) * n
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    prof = pmc.get("logsv_w_kernel", {})
    traffic = prof.get("hbm_bytes") if prof.get("config") == {"paths": n, "steps": nb_steps} else None
    return {"kernel": "logsv_w_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes, "ms_per_launch": ms,
            "path_steps_per_s": n * nb_steps / (ms * 1e-3),
            "config": {"paths": n, "steps": nb_steps, "bytes_per_path_step": 16}}


def clock_from_stamps(stamps) -> dict:
    """svmc_clock_probe_read's stamps -> the shader clock the first / last block's wave 0 saw between kernel entry and the
    end of its time loop: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz)"""
    out = {}
    for tag, (t0, r0, t1, r1) in (("first_block", stamps[0:4]), ("last_block", stamps[4:8])):
        if r1 > r0 and t1 > t0:
            out[tag] = {"mhz": 100.0 * (t1 - t0) / (r1 - r0), "wave_lifetime_ms": (r1 - r0) / 1e5}
    if out:
        out["mhz"] = float(np.mean([v["mhz"] for v in out.values()]))
    return out


def kernel_rooflines(kernel: str, k_ms: float, launches: int, n_local: int, wl, isa, pmc, clock_mhz, stamps=None) -> dict:
    """the stepping kernel of one chain call against (a) the VALU issue port -- the roof that binds it: the time loop's
    instructions by opcode class (from the loaded library's assembly) x the measured issue cost of each class, (b) HBM"""
    nb, m = wl["nb_total"], len(wl["grids"])
    hist = isa["kernels"].get(kernel, {})
    prof = pmc.get(kernel, {})
    # a chain pricing starts every path from the same constants (svmc_*_rng_from): 24 B state written per chain, 8 B terminal-x
    # snapshot per expiry, no state read
    alg_bytes = (24.0 + 8.0 * m) * n_local
    traffic = prof.get("hbm_bytes") if prof.get("config") == {"paths": n_local, "steps": nb} else None
    wave_steps = (n_local / 64.0) * nb
    out = {}
    classes = hist.get("classes")
    if classes:
        steps_per_iteration = 2.0              # one Philox call = one trip of the time loop = two time steps
        cyc = sum(CLASS_CYCLES[c] * n for c, n in classes.items()) / steps_per_iteration
        cyc_in_stream = sum((INT32_IN_FP64_STREAM_CYCLES if c == "int32" else CLASS_CYCLES[c]) * n
                            for c, n in classes.items()) / steps_per_iteration
        per_step = hist["valu"] / steps_per_iteration
        achieved = cyc * wave_steps / (k_ms * 1e-3)
        measured = prof.get("valu_insts_per_wave_step") if pmc.get("matches_loaded_library") else None
        out["roofline"] = {
            "kernel": kernel, "bound": "valu_issue", "achieved": achieved, "peak": SIMD_CYCLES_PEAK,
            "unit": "SIMD issue cycles/s", "frac": achieved / SIMD_CYCLES_PEAK,
            "frac_in_stream_int32_cost": cyc_in_stream * wave_steps / (k_ms * 1e-3) / SIMD_CYCLES_PEAK,
            "issue_cycles_per_wave_step": cyc, "insts_per_wave_step": per_step,
            "classes_per_loop_trip": classes, "class_cycles": CLASS_CYCLES, "steps_per_loop_trip": steps_per_iteration,
            "lds_insts_per_wave_step": hist.get("lds", 0) / steps_per_iteration,
            "insts_source": f"{isa['source']} (assembly of the loaded library, sha256 {isa['lib_sha256'][:16]})",
            "rates_source": "profiles/r03_valu_rates.txt (tools/ubench/valu_rates.hip)",
            "stale": bool(isa["stale"]),
            "insts_per_wave_step_counters": measured,      # SQ_INSTS_VALU of profiles/rNN_pmc.json, when it is this build's
            "peak_definition": "1024 SIMDs x 2400 MHz; a wave64 instruction holds its SIMD for the cycles of its class",
            # informational: the SMU's engine-clock reading sampled while the same call repeats for 2 s after the timed region;
            # the sensor averages and lags (boxes of the pool have reported 2100-2395 MHz for the same kernel time), so the
            # fraction above is against the 2400 MHz maximum, never against this reading
            "clock_mhz_sensor": clock_mhz, "ms_per_launch": k_ms, "launches": launches,
            "traffic": traffic, "algorithmic_bytes": alg_bytes,
        }
        # an implementation-independent reading beside the issue roofline: the fp64-class instructions of
        # the loop (FMA / add / mul / cvt / 64-bit mad: two flops each at the vector unit's rate) against the 78.6 TFLOP/s fp64
        # vector peak -- what is left of the issue budget is Philox int32 and the two quarter-rate reciprocals
        fp64_per_wave_step = classes.get("fp64", 0) / steps_per_iteration
        out["roofline"]["fp64_insts_per_wave_step"] = fp64_per_wave_step
        out["roofline"]["fp64_fma_frac"] = fp64_per_wave_step * 2.0 * 64.0 * wave_steps / (k_ms * 1e-3) / FP64_VECTOR_PEAK_FLOPS
        out["roofline"]["fp64_vector_peak_tflops"] = FP64_VECTOR_PEAK_FLOPS / 1e12
        out["roofline"]["cycles_per_wave_step_measured"] = None
        out["roofline"]["lds_busy_frac"] = None
        clk = clock_from_stamps(stamps) if stamps is not None else {}
        if clk.get("mhz"):
            # SIMD cycles the chip delivered per wave-step at the clock measured inside the kernel, beside the issue cycles the
            # loop's instructions need (issue_cycles_per_wave_step): the gap is idle issue -- LDS waits, the launch's tail
            out["roofline"]["cycles_per_wave_step_measured"] = (k_ms * 1e-3) * clk["mhz"] * 1e6 * N_SIMD / wave_steps
            # the clock MEASURED INSIDE the last timed launch (s_memtime against the 100 MHz s_memrealtime, wave 0 of the
            # launch's first and last block): the same issue cycles against the cycles the chip actually delivered in the
            # kernel's time -- what is left of 1 is idle issue, not clock
            sustained = N_SIMD * clk["mhz"] * 1e6
            out["roofline"].update({
                "clock_mhz_in_kernel": clk["mhz"], "clock_probe": clk,
                "frac_at_sustained_clock": achieved / sustained,
                "frac_in_stream_at_sustained_clock": cyc_in_stream * wave_steps / (k_ms * 1e-3) / sustained,
                "clock_note": "frac = issue cycles / (1024 SIMDs x 2400 MHz x kernel time); frac_at_sustained_clock = the same "
                              "cycles / (1024 SIMDs x clock_mhz_in_kernel x kernel time)"})
        else:
            out["roofline"].update({"clock_mhz_in_kernel": None, "frac_at_sustained_clock": None})
        if prof.get("note") and traffic is not None:
            out["roofline"]["traffic_note"] = prof["note"]
        gui, lds, conf, act = (prof.get(k + "_per_dispatch") for k in ("grbm_gui_active", "sq_lds_idx_active", "sq_lds_bank_conflict",
                                                                      "sq_active_inst_valu"))
        if pmc.get("matches_loaded_library") and gui and lds and act:
            # the second pipe the loop leans on, from the committed counter pass of THIS build: the randomly indexed table reads
            # of the inverse-CDF draw keep the LDS nearly as busy as the vector ALU
            cycles = gui / 8.0                                 # GRBM_GUI_ACTIVE counts per XCD
            out["roofline"]["lds_busy_frac"] = lds / 256.0 / cycles
            out["roofline"]["counters"] = {
                "valu_busy_frac": 4.0 * act / N_SIMD / cycles,                  # SQ_ACTIVE_INST_VALU ticks in quad-cycles
                "lds_busy_frac": lds / 256.0 / cycles,                          # SQ_LDS_IDX_ACTIVE: LDS-array cycles, per CU
                "lds_bank_conflict_share": (conf / lds) if conf else None,
                "source": f"{pmc.get('file')} (rocprofv3 --pmc, same library)"}
    hbm_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
    hbm = {"kernel": kernel, "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": hbm_gbs / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes, "ms_per_launch": k_ms, "launches": launches,
           "traffic": traffic,
           "note": "on-device-RNG stepping moves 24 + 8 M bytes per path per chain and nothing inside the time loop: "
                   "not HBM-bound by construction"}
    out["roofline_hbm"] = hbm
    if "roofline" not in out:                  # no histogram for this kernel (libsvmc.isa.json missing): report the HBM roof
        out["roofline"] = dict(hbm, stale=True)
    return out


class Watchdog:
    """fail fast instead of hanging: arm(seconds, what) gives the phase that follows a deadline; when it passes, the C-level
    timer thread of `faulthandler` (it needs no GIL, so it fires even while the main thread sits inside a HIP / RCCL call)
    dumps every thread's traceback to stderr and ends the process with status 1 -- torch.distributed.run then tears the
    other ranks down and the command returns non-zero.  A best-effort Python timer names the phase one second earlier."""

    def __init__(self, rank: int):
        self.rank, self._timer = rank, None

    def arm(self, seconds: float, what: str) -> None:
        self.disarm()
        faulthandler.dump_traceback_later(seconds, exit=True, file=sys.stderr)
        self._timer = threading.Timer(max(seconds - 1.0, 0.0), self._announce, (seconds, what))
        self._timer.daemon = True
        self._timer.start()

    def _announce(self, seconds, what):
        sys.stderr.write(f"bench.py rank {self.rank}: '{what}' did not finish within {seconds:.0f} s -- aborting this rank "
                         f"(exit status 1; traceback follows)\n")
        sys.stderr.flush()

    def disarm(self) -> None:
        faulthandler.cancel_dump_traceback_later()
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


def max_rel_dev(got, ref) -> float:
    """largest |got - ref| / |ref| over the entries of two lists of arrays; an entry with ref == 0 counts as 0 when got is
    0 too and as inf otherwise (nothing is silently dropped)"""
    worst = 0.0
    for a, b in zip(got, ref):
        a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
        if a.shape != b.shape:
            return float("inf")
        nz = b != 0.0
        if np.any(nz):
            worst = max(worst, float(np.nanmax(np.abs(a[nz] / b[nz] - 1.0))))
        if np.any(~nz) and np.any(a[~nz] != 0.0):
            return float("inf")
        if np.any(np.isnan(a) != np.isnan(b)):
            return float("inf")
    return worst


class CAbiChain:
    """the workload through the fused C driver of the boundary: ONE svmc_logsv_chain_price call per chain on a
    svmc_session_t (include/svmc.h; what examples/price_chain.c and price_chain_rccl.c call) -- host arrays in, prices
    out, every launch queued inside the library.  With `comm` (an svmc_comm_t) the session holds this rank's shard."""

    def __init__(self, svlib, wl, P, n_local: int, comm_handle=None, rank: int = 0, world: int = 1, n_total: int = 0,
                 offset: int = 0):
        import ctypes as C
        self.C, self.svlib, self.L = C, svlib, svlib.load()
        self.P, self.wl = P, wl
        m = len(wl["ttms"])
        f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)                                    # noqa: E731
        self.ttms, self.fw, self.df, self.eta = f64(wl["ttms"]), f64(wl["forwards"]), f64(wl["dfs"]), np.ones(m)
        self.kk = f64(np.concatenate(wl["strikes"]))
        codes = {"C": 0, "P": 1, "IC": 2, "IP": 3}
        self.codes = np.ascontiguousarray([codes[str(t)] for t in np.concatenate(wl["types"])], dtype=np.int8)
        self.offs = np.concatenate([[0], np.cumsum([len(k) for k in wl["strikes"]])]).astype(np.uintp)
        self.counts = [len(k) for k in wl["strikes"]]
        self.prices, self.errs = np.empty(self.kk.size), np.empty(self.kk.size)
        self.sess = C.c_void_p()
        svlib.check(self.L.svmc_session_create(C.byref(self.sess), n_local, m, int(self.kk.size)))
        if comm_handle is not None:
            svlib.check(self.L.svmc_session_set_comm(self.sess, comm_handle, rank, world, n_total, offset))
        dp = C.POINTER(C.c_double)
        self._args = (self.ttms.ctypes.data_as(dp), self.fw.ctypes.data_as(dp), self.df.ctypes.data_as(dp),
                      self.eta.ctypes.data_as(dp), m, self.kk.ctypes.data_as(dp),
                      self.codes.ctypes.data_as(C.POINTER(C.c_int8)), self.offs.ctypes.data_as(C.POINTER(C.c_size_t)))
        self._out = (self.prices.ctypes.data_as(dp), self.errs.ctypes.data_as(dp))

    def price(self, seed: int):
        P = self.P
        self.svlib.check(self.L.svmc_logsv_chain_price(self.sess, *self._args, P.sigma0, P.theta, P.kappa1, P.kappa2, P.beta,
                                                       P.volvol, 1, self.wl["spy"], 1, seed, 0, *self._out))
        return np.split(self.prices.copy(), np.cumsum(self.counts)[:-1]), np.split(self.errs.copy(), np.cumsum(self.counts)[:-1])

    def close(self):
        if self.sess is not None:
            self.L.svmc_session_destroy(self.sess)
            self.sess = None


def c_abi_route_leg(make_chain, python_step, calls: int, n_paths_job: int, nb: int, barrier, max_over_ranks) -> dict:
    """`calls` chains through CAbiChain.price, each timed on the host clock and INTERLEAVED call by call with the Python route on
    the same seed (the chip's clock drifts over a run: two routes timed in separate windows differ by the drift, not by the
    route -- tools/ubench/fused_driver_overhead.py), prices compared on every call.  Runs with the garbage collector frozen
    (main() did that before its own timed region)."""
    chain = make_chain()
    try:
        for i in range(3):
            chain.price(7 + i)
        barrier()
        t_c, t_py, equal, worst = [], [], True, 0.0
        for i in range(calls):
            t0 = time.perf_counter()
            p_py, e_py = python_step(i)
            t1 = time.perf_counter()
            p_c, e_c = chain.price(20240602 + i)
            t2 = time.perf_counter()
            t_py.append(t1 - t0)
            t_c.append(t2 - t1)
            equal = equal and all(np.array_equal(a, b) for a, b in zip(p_c, p_py)) and \
                all(np.array_equal(a, b) for a, b in zip(e_c, e_py))
            worst = max(worst, max_rel_dev(p_c, p_py))
        barrier()
        t_c, t_py = np.array(t_c), np.array(t_py)
        med = max_over_ranks(float(np.median(t_c)))
        med_py = max_over_ranks(float(np.median(t_py)))
        return {"entry_point": "svmc_logsv_chain_price (one C-ABI call per chain, svmc_session_t)", "calls": calls,
                "ms_per_step": 1e3 * med, "ms_per_step_max": 1e3 * max_over_ranks(float(t_c.max())),
                "ms_per_step_mean": 1e3 * max_over_ranks(float(t_c.mean())), "value": n_paths_job * nb / med,
                "python_route_ms_per_step_interleaved": 1e3 * med_py, "c_over_python_interleaved": med / med_py,
                "prices_equal_python_route": bool(equal), "max_rel_dev_vs_python_route": worst}
    finally:
        chain.close()


# ---------------------------------------------------------------------------------------------------------------
# `secondary`: the other BASELINE configurations, compact, each with a parity scalar against the oracle (N = 1 only)
# ---------------------------------------------------------------------------------------------------------------
C5_SETS = (("btc", None), ("readme", dict(sigma0=0.8327, theta=1.0139, kappa1=4.8609, kappa2=4.7940, beta=0.1988, volvol=2.3694)),
           ("quick", dict(sigma0=1.0, theta=1.0, kappa1=5.0, kappa2=5.0, beta=0.2, volvol=2.0)),
           ("test", dict(sigma0=0.2, theta=0.22, kappa1=3.0, kappa2=12.0, beta=-0.3, volvol=0.4)),
           ("fig3", dict(sigma0=1.5, theta=1.0, kappa1=4.0, kappa2=4.0, beta=0.0, volvol=1.5)))


def _sig(v, digits=4):
    return float(f"{float(v):.{digits}g}")


def _median_ms(fn, reps, warm):
    import ctypes as C  # noqa: F401
    from stochvolmodels_amd import _lib
    sync = lambda: _lib.load().svmc_stream_synchronize(None)          # noqa: E731
    for _ in range(warm):
        fn()
    sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        sync()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), out


def _prices_dev(got, ref, floor):
    """largest |gpu - oracle| / (|oracle| + floor) over a chain's prices"""
    g, o = np.concatenate([np.ravel(a) for a in got]), np.concatenate([np.ravel(a) for a in ref])
    return float(np.max(np.abs(g - o) / (np.abs(o) + floor)))


def secondary_block(sv, P) -> dict:
    """BASELINE configs 1, 3, 5 and the rows the default line does not time, in the line's own process.  Every
    figure is the median wall time of whole product calls; every `dev` is the largest deviation of the GPU's PRICES from the
    oracle's on the SAME counter-based stream at a reduced path count (the full-size comparisons are tests/test_gpu_fullsize.py).
    Compact on purpose: the block is the END of the line, the part a truncating log keeps."""
    from oracle import oracle                              # the checker of the parity scalars, never the thing timed
    from stochvolmodels_amd.engine import get_engine
    oracle.build()
    oracle.set_threads(oracle.effective_cores())
    t_start = time.perf_counter()
    out = {"note": "ms: median wall of one product call; psps: path-steps/s; dev: max rel price deviation from the oracle, same "
                   "stream, n_dev paths"}
    n_dev = 1 << 14
    kk = np.linspace(0.5, 1.5, 21)
    types = np.where(kk >= 1.0, "C", "P")

    # ---- C1: Heston Euler, 10 000 paths x 100 steps, 5 strikes (the reference's CPU-runnable case)
    k5, t5 = np.array([0.8, 0.9, 1.0, 1.1, 1.2]), np.array(["P", "P", "C", "C", "C"])
    h0 = dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4)
    c1 = dict(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1), strikes_ttms=(k5,), optiontypes_ttms=(t5,))
    ms, (pr, _) = _median_ms(lambda: sv.heston_mc_chain_pricer(nb_path=10_000, nb_steps_per_year=99, seed=20240601, **c1, **h0), 20, 3)
    x, v, q = oracle.heston_terminal_rng(np.zeros(10_000), h0["v0"] * np.ones(10_000), np.zeros(10_000), 100, 0.01, h0["theta"],
                                         h0["kappa"], h0["rho"], h0["volvol"], 20240601)
    opr, _ = oracle.payoff(x, q, 1.0, 1.0, k5, t5)
    out["c1"] = {"ms": _sig(ms), "psps": _sig(1e6 / (ms * 1e-3)), "dev": _sig(_prices_dev(pr, [opr], 1e-3), 2)}

    # ---- C3: Heston, 2^22 paths x 4 x 128 steps, 4 x 21 strikes; Euler (the reference's scheme) and QE, both parameter sets
    ttms = np.array([0.25, 0.5, 0.75, 1.0])
    chain4 = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4)
    hb = sv.BTC_HESTON_PARAMS
    c3 = {"paths": 1 << 22, "steps": 512, "n_dev": n_dev}
    for tag, par in (("base", h0), ("btc", dict(v0=hb.v0, theta=hb.theta, kappa=hb.kappa, rho=hb.rho, volvol=hb.volvol))):
        for scheme in ("euler", "qe"):
            ms, _ = _median_ms(lambda: sv.heston_mc_chain_pricer(nb_path=1 << 22, scheme=scheme, nb_steps_per_year=508, seed=20240603,
                                                                 **chain4, **par), 3, 1)
            pr, _ = sv.heston_mc_chain_pricer(nb_path=n_dev, scheme=scheme, nb_steps_per_year=508, seed=20240603, **chain4, **par)
            x, v, q = np.zeros(n_dev), par["v0"] * np.ones(n_dev), np.zeros(n_dev)
            ref, step0 = [], 0
            for ttm in ttms:
                x, v, q = oracle.heston_terminal_rng(x, v, q, 128, 0.25 / 128, par["theta"], par["kappa"], par["rho"], par["volvol"],
                                                     20240603, scheme=oracle.HESTON_QE if scheme == "qe" else oracle.HESTON_EULER_FLOOR,
                                                     step_offset=step0)
                ref.append(oracle.payoff(x, q, float(ttm), 1.0, kk, types)[0])
                step0 += 128
            c3[f"{scheme}_{tag}"] = {"ms": _sig(ms), "psps": _sig((1 << 22) * 512 / (ms * 1e-3)), "dev": _sig(_prices_dev(pr, ref, 1e-3), 2)}
    out["c3"] = c3

    # ---- C5: five parameter sets; analytic chain of all five in one batch; 2^23-path Monte Carlo per set; the reference's verdict
    t5_ = np.arange(1, 5) / 8.0                            # C4's first four expiries: ttm = k / 8
    fw = 67000.0 * np.exp(0.05 * t5_)
    strikes5 = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types5 = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes5, fw))
    chain5 = dict(ttms=t5_, forwards=fw, discfactors=np.exp(-0.05 * t5_), strikes_ttms=strikes5, optiontypes_ttms=types5)
    sets = [P if v is None else sv.LogSvParams(**v) for _, v in C5_SETS]
    oc = sv.OptionChain(ttms=t5_, forwards=fw, strikes_ttms=strikes5, optiontypes_ttms=types5, ids=None, discfactors=chain5["discfactors"])
    pricer = sv.LogSVPricer()
    ms_an, an = _median_ms(lambda: pricer.price_chain_batch(oc, sets), 5, 1)
    c5 = {"paths": 1 << 23, "steps": 512, "sets": [t for t, _ in C5_SETS], "analytic_batch_ms": _sig(ms_an), "mc_ms": [], "dev": [],
          "pass_of_84": [], "n_dev": n_dev}
    for p_, a_ in zip(sets, an):
        kw = dict(v0=p_.sigma0, theta=p_.theta, kappa1=p_.kappa1, kappa2=p_.kappa2, beta=p_.beta, volvol=p_.volvol,
                  vol_backbone_etas=np.ones(4), nb_steps_per_year=1016, seed=20240610, **chain5)
        ms, (pr, sd) = _median_ms(lambda: sv.logsv_mc_chain_pricer(nb_path=1 << 23, **kw), 2, 1)
        z = (np.stack(pr) - np.stack(a_)) / np.where(np.stack(sd) > 0, np.stack(sd), np.nan)
        c5["mc_ms"].append(_sig(ms))
        c5["pass_of_84"].append(int(np.sum(np.abs(z) <= 4.0)))     # |analytic - MC| <= 4 stderr, the reference's criterion
        pr, _ = sv.logsv_mc_chain_pricer(nb_path=n_dev, **kw)
        x, s_, q = np.zeros(n_dev), p_.sigma0 * np.ones(n_dev), np.zeros(n_dev)
        ref, step0 = [], 0
        for i, ttm in enumerate(t5_):
            x, s_, q = oracle.logsv_terminal_rng(x, s_, q, 128, 0.125 / 128, p_.theta, p_.kappa1, p_.kappa2, p_.beta, p_.volvol,
                                                 20240610, step_offset=step0)
            ref.append(oracle.payoff(x, q, float(ttm), float(fw[i]), strikes5[i], types5[i], float(chain5["discfactors"][i]))[0])
            step0 += 128
        c5["dev"].append(_sig(_prices_dev(pr, ref, 1e-3 * float(fw[0])), 2))
    c5["mc_psps"] = _sig((1 << 23) * 512 / (np.mean(c5["mc_ms"]) * 1e-3))
    p0 = sets[0]
    o_an = oracle.logsv_chain_pricer((p0.sigma0, p0.theta, p0.kappa1, p0.kappa2, p0.beta, p0.volvol), t5_, fw, chain5["discfactors"],
                                     strikes5, types5)
    c5["analytic_dev_btc"] = _sig(_prices_dev(an[0], o_an, 1e-3 * float(fw[0])), 2)
    out["c5"] = c5

    # ---- f3: one calibration objective evaluation on FROZEN randoms (nothing resident), 4 x 13 chain, 10^5 paths x 364 steps
    tt = np.array([1 / 12, 0.25, 0.5, 1.0])
    k13 = np.linspace(0.7, 1.3, 13)
    ch = dict(ttms=tt, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k13,) * 4, optiontypes_ttms=(np.where(k13 >= 1.0, "C", "P"),) * 4)
    res = sv.draw_fixed_randoms_on_device(tt, nb_path=100_000, nb_steps_per_year=360, seed=10)
    bumped = [sv.LogSvParams(sigma0=P.sigma0 + 1e-3 * j, theta=P.theta, kappa1=P.kappa1 + 1e-2 * j, kappa2=P.kappa2, beta=P.beta,
                             volvol=P.volvol - 1e-2 * j) for j in range(7)]
    one = lambda: sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, v0=P.sigma0, theta=P.theta, kappa1=P.kappa1,  # noqa: E731
                                                         kappa2=P.kappa2, beta=P.beta, volvol=P.volvol, vol_backbone_etas=np.ones(4),
                                                         return_ivols=True, **ch)
    ms1, got = _median_ms(one, 50, 3)
    ms7, _ = _median_ms(lambda: sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=bumped, W0s=res, return_ivols=True, **ch), 30, 3)
    want = sv.logsv_mc_chain_pricer(v0=P.sigma0, theta=P.theta, kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta, volvol=P.volvol,
                                    vol_backbone_etas=np.ones(4), nb_path=100_000, nb_steps_per_year=360, seed=10, **ch)
    res.free()
    out["f3_frozen"] = {"paths": 100_000, "steps": 364, "one_set_with_ivols_ms": _sig(ms1), "seven_sets_with_ivols_ms": _sig(ms7),
                        "hbm_bytes_for_randoms": 0,
                        "bit_equal_to_mc_chain_pricer": bool(all(np.array_equal(a, b) for a, b in zip(got[0] + got[1], want[0] + want[1])))}

    # ---- mid: the reference's own path counts (10^5 by default, pricers/logsv_pricer.py:374; 4 x 10^5 in its paper), 4 x 13 chain x 364
    #      steps -- 1.5 and 6.1 waves per SIMD: the few-waves (pipelined) generators, one C-ABI call per chain
    mid = {"chain": "4 x 13 strikes, ttm 1/12 .. 1, 364 steps", "n_dev": n_dev}
    for n_mid in (100_000, 400_000):
        kw = dict(v0=P.sigma0, theta=P.theta, kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta, volvol=P.volvol, vol_backbone_etas=np.ones(4),
                  nb_steps_per_year=360, seed=20240611, **ch)
        ms, _ = _median_ms(lambda: sv.logsv_mc_chain_pricer(nb_path=n_mid, **kw), 60, 5)
        mid[str(n_mid)] = {"ms": _sig(ms), "psps": _sig(n_mid * 364 / (ms * 1e-3))}
    pr, _ = sv.logsv_mc_chain_pricer(nb_path=n_dev, **kw)
    x, s_, q = np.zeros(n_dev), P.sigma0 * np.ones(n_dev), np.zeros(n_dev)
    ref, step0, t0 = [], 0, 0.0
    for ttm in tt:
        nb_i, dt_i, _ = sv.set_time_grid(float(ttm) - t0, 360)
        x, s_, q = oracle.logsv_terminal_rng(x, s_, q, nb_i, dt_i, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 20240611, step_offset=step0)
        ref.append(oracle.payoff(x, q, float(ttm), 1.0, k13, np.where(k13 >= 1.0, "C", "P"))[0])
        step0, t0 = step0 + nb_i, float(ttm)
    mid["dev"] = _sig(_prices_dev(pr, ref, 1e-3), 2)
    out["mid"] = mid

    # ---- C2's call at 2^21 paths: the asymptotic rate (the 2^20-path launch ends in a second, partial residency round)
    wl = dict(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1), strikes_ttms=(kk,), optiontypes_ttms=(types,))
    ms, _ = _median_ms(lambda: sv.logsv_mc_chain_pricer(v0=P.sigma0, theta=P.theta, kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta,
                                                        volvol=P.volvol, vol_backbone_etas=np.ones(1), nb_path=1 << 21,
                                                        nb_steps_per_year=1023, seed=20240602, **wl), 15, 10)   # (warm: the block's small launches before it leave the clocks low)
    out["c2_at_2e21_paths"] = {"ms": _sig(ms), "psps": _sig((1 << 21) * 1024 / (ms * 1e-3))}
    for n in (1 << 22, 1 << 23, 1 << 21, 100_000, 400_000, n_dev, 10_000):            # give the legs' HBM back
        try:
            get_engine(n).close()
        except Exception:                                   # noqa: BLE001
            pass
    out["seconds"] = _sig(time.perf_counter() - t_start, 3)
    return out


# ---------------------------------------------------------------------------------------------------------------
# one process, N devices (svmc_multi_*)
# ---------------------------------------------------------------------------------------------------------------
def probe_single_process_rccl(devices, timeout: float):
    """can ncclCommInitAll + one all-reduce over `devices` come up in ONE process?  Asked of a child process under a deadline
    (an RCCL initialisation that hangs cannot be interrupted from inside): -> (ok, reason)"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from stochvolmodels_amd.multi import MultiDeviceSession\n"
            "ms = MultiDeviceSession(%r, 1 << 16, 1, 8, reduce='rccl'); i = ms.info(); ms.close()\n"
            "assert i['reduce'] == 'rccl' and i['rccl_ranks_seen'] == %d, i\n" % (ROOT, list(devices), len(devices)))
    if os.environ.get("SVMC_BENCH_FAULT") == "multi_rccl_init":
        return False, "fault injection: the single-process RCCL probe fails"
    try:
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout,
                             env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    except subprocess.TimeoutExpired:
        return False, f"probe did not finish within {timeout:.0f} s (killed)"
    if run.returncode == 0:
        return True, ""
    tail = [ln for ln in (run.stdout + run.stderr).strip().splitlines() if ln.strip()]
    return False, (tail[-1] if tail else f"exit status {run.returncode}")[:240]


def single_process_leg(args, cfg: str, per_gpu: int, world: int, n_devices: int, steps: int, timeout: float) -> dict:
    """rank 0 of a launched N > 1 run starts `bench.py --gpus N --single-process` as a CHILD (its own process: nothing it does --
    an RCCL initialisation that hangs included -- can take the launched run down) on the same job and returns its line's essentials"""
    devices = [r % max(n_devices, 1) for r in range(world)]
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--steps", str(steps), "--warmup", "1",
           "--config", cfg, "--paths-per-gpu", str(per_gpu), "--devices", ",".join(str(d) for d in devices)]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                        "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "SVMC_DIST_BACKEND", "SVMC_BENCH_SELF_LAUNCHED")
           and not k.startswith("TORCHELASTIC")}
    env["SVMC_BENCH_PREWARM"] = "2"
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"the single-process child did not finish within {timeout:.0f} s (killed)"}
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": ("exit status %d: " % run.returncode) + (run.stderr.strip().splitlines() or ["no output"])[-1][:240]}
    out = json.loads(lines[-1])
    out["child_exit_status"] = run.returncode
    return out


def single_process_main(args) -> int:
    """`python bench.py --gpus N --single-process`: the C4 (or --config) job of N x paths-per-gpu paths by ONE process driving N
    devices through a multi-session -- no launcher, no rendezvous.  Same timed region as the launched form (W warm-up calls, K
    timed calls between synchronisations); prints one JSON line."""
    import ctypes as C

    import stochvolmodels_amd as sv
    from stochvolmodels_amd import _lib as svlib
    from stochvolmodels_amd.multi import get_multi_session
    wd = Watchdog(0)
    n_dev = C.c_int(0)
    svlib.check(svlib.load().svmc_device_count(C.byref(n_dev)))
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit(f"--devices names {len(devices)} devices, --gpus {args.gpus}")
    if max(devices) >= n_dev.value:
        raise SystemExit(f"bench.py --single-process --gpus {args.gpus}: only {n_dev.value} GPU(s) visible (--devices 0,0,.. lets "
                         f"shards share a device for testing)")
    P = sv.LOGSV_BTC_PARAMS
    cfg = args.config or ("c2" if args.gpus == 1 else "c4")
    wl = make_workload(cfg, sv)
    per_gpu = args.paths_per_gpu or ((1 << 20) if cfg == "c2" else (1 << 21))
    n_total, nb = per_gpu * args.gpus, wl["nb_total"]
    reduce, why = args.reduce, None
    if reduce == "auto":
        if len(set(devices)) < len(devices):
            reduce, why = "host", "shards share a device (RCCL takes one rank per device)"
        else:
            ok, why = probe_single_process_rccl(devices, float(os.environ.get("SVMC_BENCH_INIT_TIMEOUT", "120")))
            reduce = "rccl" if ok else "host"
    wd.arm(float(os.environ.get("SVMC_BENCH_LEG_TIMEOUT", "600")), "single-process multi-device run")

    def step(i):
        return price(sv, wl, P, n_total, 20240602 + i, devices=devices, reduce=reduce)

    step(-2000)
    gc.collect()
    gc.freeze()
    for i in range(PREWARM + args.warmup):
        step(-1000 - i)
    t0 = time.perf_counter()
    for i in range(args.steps):
        prices, stderrs = step(i)
    elapsed = time.perf_counter() - t0
    info = get_multi_session(devices, n_total, len(wl["ttms"]), wl["n_strikes"], reduce=reduce).info()
    p777, _ = step(777)
    wd.disarm()
    line = {"metric": "MC path-steps/sec", "value": float(n_total) * nb * args.steps / elapsed, "unit": "path-steps/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["label"], "paths_per_gpu": per_gpu, "paths_total": n_total, "time_steps": nb,
                       "expiries": len(wl["grids"]), "strikes": wl["n_strikes"],
                       "parallelism": f"path-sharded x{args.gpus}, one process (svmc_multi_*)"},
            "comm": "single-process multi-session", "reduce": info["reduce"], "reduce_fallback_reason": why or None,
            "rccl_ranks_seen": info["rccl_ranks_seen"] or None, "shards_agree_bitwise": info["shards_agree"],
            "devices": devices, "shard_ms_last_call": [round(s_["last_call_ms"], 3) for s_ in info["shards"]],
            "prices_seed_777": [float(v) for v in np.concatenate(p777)],
            "prices_head": [float(v) for v in prices[0][:3]], "stderr_head": [float(v) for v in stderrs[0][:3]]}
    print(json.dumps(line), flush=True)
    return 0 if info["shards_agree"] else 3


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, the way the driver's
    own command does -- one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 -- and pass their output
    through (rank 0 prints the line).  The launch runs under an overall deadline (SVMC_BENCH_TIMEOUT, 1500 s -- below the
    driver's own 1800 s): past it the whole process group is killed and the command returns 124 instead of hanging."""
    import ctypes as C
    import signal

    from stochvolmodels_amd import _lib
    count = C.c_int(0)
    _lib.check(_lib.load().svmc_device_count(C.byref(count)))
    # test modes: the ranks share the GPUs there are -- straight on the gloo rung, or through the whole fallback ladder
    sharing = os.environ.get("SVMC_DIST_BACKEND") == "gloo" or os.environ.get("SVMC_BENCH_SHARE_DEVICES") == "1"
    if count.value < args.gpus and not sharing:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {count.value} GPU(s) visible -- one rank per GPU is required "
                         f"(RCCL refuses two ranks on one device; SVMC_BENCH_SHARE_DEVICES=1 lets ranks share a GPU for testing: "
                         f"the ladder then lands on its gloo rung)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SVMC_BENCH_SELF_LAUNCHED="1")
    limit = float(os.environ.get("SVMC_BENCH_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)

    def kill_ranks():
        try:
            os.killpg(proc.pid, signal.SIGKILL)        # the launcher and every rank it started (its own session)
        except ProcessLookupError:
            pass
        proc.wait()

    # the ranks live in a session of their own (so that the deadline can end them all at once): a Ctrl-C or SIGTERM that
    # reaches only THIS process must take them along, or they stay on the GPUs as orphans
    def on_term(signum, frame):
        raise SystemExit(128 + signum)

    old_term = signal.signal(signal.SIGTERM, on_term)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        sys.stderr.write(f"bench.py --gpus {args.gpus}: the ranks did not finish within {limit:.0f} s -- killing them\n")
        kill_ranks()
        return 124
    except (KeyboardInterrupt, SystemExit):
        kill_ranks()
        raise
    finally:
        signal.signal(signal.SIGTERM, old_term)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.single_process:
        if world > 1:
            raise SystemExit("--single-process is ONE process: run it without a launcher")
        sys.exit(single_process_main(args))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        if os.environ.get("SVMC_BENCH_SELF_LAUNCHED") == "1":
            raise SystemExit("bench.py: launched ranks without WORLD_SIZE (torch.distributed.run did not set the environment)")
        sys.exit(self_launch(args))

    import torch  # device plumbing + torch.distributed only
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import dist as svdist
    from stochvolmodels_amd.engine import get_engine

    # deadlines (seconds): every rank must reach the rendezvous (the slowest rank's `import torch` on a cold box is inside
    # this one), the collective layer must initialise, and each leg of the run must end
    wd = Watchdog(rank)
    t_rdv = float(os.environ.get("SVMC_BENCH_RENDEZVOUS_TIMEOUT", "300"))
    t_init = float(os.environ.get("SVMC_BENCH_INIT_TIMEOUT", "120"))
    t_leg = float(os.environ.get("SVMC_BENCH_LEG_TIMEOUT", "600"))
    if os.environ.get("SVMC_BENCH_FAULT") == f"hang_init:{rank}":        # test hook: this rank never reaches the rendezvous
        wd.arm(t_init, "fault injection: hang before the rendezvous")
        time.sleep(10 * t_init + 60)

    def on_phase(name):
        if name == "rendezvous":
            wd.arm(t_rdv, "rendezvous of the ranks (torch.distributed.init_process_group)")
        elif name == "collective_probe":
            wd.arm(t_init + 60.0, "probe of a collective rung in a child process (the child has its own deadline)")
        elif name == "collective_init":
            wd.arm(t_init, "initialisation of the collective layer (RCCL communicator + first all-reduce)")
        else:
            wd.disarm()

    ladder = None
    if world > 1:
        # the collective layer by the fallback ladder: gloo control plane, then torch-nccl -> libsvmc RCCL -> gloo, each probed in
        # a child process first (SVMC_DIST_BACKEND=gloo, the ranks-share-a-GPU test mode, asks for the last rung outright)
        rungs = ("gloo",) if os.environ.get("SVMC_DIST_BACKEND") == "gloo" else ("nccl", "rccl", "gloo")
        comm, ladder = svdist.init_with_fallback(rungs=rungs, on_phase=on_phase, probe_timeout=t_init)
        backend = ladder["backend"]
    else:
        comm = svdist.init_from_env()            # SVMC_DIST_SINGLE_RANK_GROUP=1: a lone rank with a real process group
        torch.cuda.set_device(0)
        backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    grouped = torch.distributed.is_initialized()
    # the bench's own bookkeeping collectives (barriers, maxima) run on the DEFAULT group: the gloo control plane at N > 1
    coll_dev = "cuda" if (grouped and torch.distributed.get_backend() == "nccl") else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v: float) -> float:
        if not grouped:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=coll_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def gather_over_ranks(v: float) -> list:
        if not grouped:
            return [float(v)]
        t = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        t[rank] = v
        torch.distributed.all_reduce(t)
        return [float(x) for x in t.cpu().tolist()]

    def all_ranks_ok(ok: bool) -> bool:
        """agree on success across the ranks: a leg that failed on SOME ranks must be abandoned by ALL of them, or the
        failing ranks wait in a barrier while the others wait inside a collective"""
        if not grouped:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=coll_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    P = sv.LOGSV_BTC_PARAMS
    cfg = args.config or ("c2" if world == 1 else "c4")
    wl = make_workload(cfg, sv)
    per_gpu = args.paths_per_gpu or ((1 << 20) if cfg == "c2" else (1 << 21))
    n_total = per_gpu * world
    nb = wl["nb_total"]
    kernel = "logsv_rng_kernel" if len(wl["grids"]) == 1 else "logsv_chain_rng_kernel"
    from stochvolmodels_amd import _lib as svlib
    isa = load_isa(svlib.LIB_PATH)
    pmc = load_pmc(isa["lib_sha256"])

    def step(i):
        return price(sv, wl, P, n_total, 20240602 + i)

    wd.arm(t_leg, "first call + warm-up")
    step(-2000)                                # first call: library load, buffers, first launches
    # no full garbage collection inside the timed region: everything alive now (the imports' ~70 000 container objects)
    # moves to the permanent generation, so the collector's passes over what the steps allocate stay in the microseconds.
    # Done BEFORE the warm-up steps: the collection itself is ~60 ms of host time with the GPU idle, and the clock would
    # ramp down again behind it
    gc.collect()
    gc.freeze()
    for i in range(PREWARM):
        step(-1000 - i)
    for i in range(args.warmup):
        step(-1 - i)
    offset, n_local = svdist.shard_range(n_total, comm.rank, comm.world)
    eng = get_engine(n_local, path_offset=offset)
    wd.arm(t_leg, "timed region")
    svlib.check(svlib.load().svmc_clock_probe_arm(1))     # this thread's stepping launches stamp the shader clock (off otherwise)
    barrier()
    if os.environ.get("SVMC_BENCH_NO_KERNEL_EVENTS") != "1":     # diagnostics: the HIP events around the stepping launches
        eng.start_kernel_timing()
    t0 = time.perf_counter()
    step_ends = []
    for i in range(args.steps):
        prices, stderrs = step(i)
        step_ends.append(time.perf_counter())
    t_b = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    final_barrier_ms = 1e3 * (time.perf_counter() - t_b)
    per_step_ms = 1e3 * np.diff(np.array([t0] + step_ends))
    kernel_ms = (eng.stop_kernel_timing() if eng._prof is not None else {}).get(kernel, [float("nan")])
    # the shader clock the LAST timed launch ran at, stamped inside the kernel (svmc_clock_probe_read)
    import ctypes as C
    stamps = (C.c_uint64 * 8)()
    svlib.check(svlib.load().svmc_clock_probe_read(stamps, eng.stream))
    svlib.check(svlib.load().svmc_clock_probe_arm(0))
    elapsed = max_over_ranks(elapsed)
    value = float(n_total) * nb * args.steps / elapsed
    k_ms_all = gather_over_ranks(float(np.mean(kernel_ms)))

    extra = {}
    failures = []
    single = svdist.SingleComm()
    if world > 1 and isinstance(comm, svdist.TorchComm):
        # the stream-ordering shortcut of TorchComm (no host synchronisation around the all-reduces) against the
        # host-synchronised fallback, with real peers: the bits must be identical
        wd.arm(t_leg, "stream-ordered vs host-synchronised collectives")
        os.environ["SVMC_DIST_STRICT_SYNC"] = "1"
        p_strict, _ = step(424242)
        os.environ["SVMC_DIST_STRICT_SYNC"] = "0"
        p_ordered, _ = step(424242)
        extra["stream_ordered_equals_strict_sync"] = bool(all(np.array_equal(a, b) for a, b in zip(p_strict, p_ordered)))
    extra["comm"] = type(comm).__name__
    extra["backend"] = backend
    extra["kernel_ms_over_ranks"] = {"min": min(k_ms_all), "max": max(k_ms_all), "per_rank": [round(v, 4) for v in k_ms_all]}
    if grouped:
        # the ranks the process group itself sees (a sum all-reduce of ones): must be --gpus on any backend
        t = torch.ones(1, dtype=torch.float64, device=coll_dev)
        torch.distributed.all_reduce(t)
        extra["group_ranks_seen"] = int(round(float(t.item())))
        if extra["group_ranks_seen"] != world:
            failures.append(f"group_ranks_seen {extra['group_ranks_seen']} != --gpus {world}")
    extra["rccl_ranks_seen"] = None
    if isinstance(comm, svdist.RcclComm):
        extra["rccl_ranks_seen"] = comm.ranks_seen()
    elif world > 1 and isinstance(comm, svdist.TorchComm) and backend == "nccl":
        # the ranks the DATA plane sees: a sum all-reduce of ones through the nccl group the chains' collectives run on
        t = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        torch.distributed.all_reduce(t, group=comm.group)
        extra["rccl_ranks_seen"] = int(round(float(t.item())))
        if extra["rccl_ranks_seen"] != world:
            failures.append(f"the nccl group saw {extra['rccl_ranks_seen']} of {world} ranks")

    # ---- the fused C driver of the boundary on the line's own workload (single GPU: here; N > 1: inside the RCCL leg)
    if world == 1 and not args.no_c_abi_route:
        wd.arm(t_leg, "c_abi_route")
        extra["c_abi_route"] = c_abi_route_leg(lambda: CAbiChain(svlib, wl, P, n_local), step, max(3, min(args.steps, 20)),
                                               n_total, nb, barrier, max_over_ranks)

    dev_count = C.c_int(0)
    svlib.check(svlib.load().svmc_device_count(C.byref(dev_count)))
    ranks_share_a_device = world > dev_count.value
    want_rccl_leg = grouped and not isinstance(comm, svdist.RcclComm) and (backend == "nccl" or not args.no_extra_legs)
    rccl_probe_failed = None
    if want_rccl_leg and ladder is not None and not ranks_share_a_device:
        tried = {p_["rung"]: p_ for p_ in ladder["probes"]}
        if "rccl" in tried and not tried["rccl"]["ok"]:
            rccl_probe_failed = tried["rccl"]["reason"]                # the ladder already tried this rung and it failed
        elif "rccl" not in tried:
            # the nccl rung held: libsvmc's own RCCL route has not been tried on this node yet -- probe it in child processes
            # before the ranks themselves create a second communicator
            wd.arm(t_init + 60.0, "probe of libsvmc's RCCL route")
            box = [free_port() if rank == 0 else None]
            torch.distributed.broadcast_object_list(box, src=0)
            ok, why = svdist._run_probe("rccl", int(box[0]), t_init)
            if not all_ranks_ok(ok):
                rccl_probe_failed = why or "the probe failed on another rank"
    if want_rccl_leg and rccl_probe_failed is not None:
        extra["rccl_route"] = {"skipped": "probe failed: " + str(rccl_probe_failed)[:200]}
    elif want_rccl_leg and ranks_share_a_device:
        # the gloo test mode (several ranks on one GPU): RCCL refuses two ranks per device (ncclCommInitRank: invalid
        # usage; tests/test_gpu_parity.py::test_two_rccl_ranks_on_one_gpu records the refusal) -- nothing to time
        extra["rccl_route"] = {"skipped": f"{world} ranks share {dev_count.value} device(s): RCCL needs one device per rank"}
    elif want_rccl_leg:
        # the same chain through libsvmc's OWN RCCL entry points (include/svmc.h svmc_rccl_*: the route a C / C++ host
        # takes; dist.RcclComm drives it from Python): a second communicator built from a unique id that the torch group
        # ships, the two all-reduces issued by libsvmc on the engine's stream.  ncclCommCount says how many ranks RCCL
        # itself sees.  Every step that can fail on SOME ranks is followed by an agreement across ALL ranks.
        wd.arm(t_init + t_leg, "rccl_route (svmc_rccl_comm_create + timed chains)")
        rc, err = None, None
        try:
            box = [svdist.RcclComm.unique_id() if rank == 0 else None]
            torch.distributed.broadcast_object_list(box, src=0)
            rc = svdist.RcclComm(rank, world, box[0])
        except Exception as exc:                             # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"[:300]
        if not all_ranks_ok(err is None):
            extra["rccl_route"] = {"error": err or "RcclComm creation failed on another rank"}
            if rc is not None:
                rc.close()
        else:
            extra["rccl_ranks_seen"] = rc.ranks_seen()
            try:
                k = max(3, min(args.steps, 20))
                for i in range(3):
                    price(sv, wl, P, n_total, 7 + i, comm=rc)
                barrier()
                t0r = time.perf_counter()
                for i in range(k):
                    p_rccl, _ = price(sv, wl, P, n_total, 20240602 + i, comm=rc)
                barrier()
                t_rccl = (time.perf_counter() - t0r) / k
            except Exception as exc:                         # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"[:300]
            if not all_ranks_ok(err is None):
                extra["rccl_route"] = {"error": err or "the RcclComm chains failed on another rank"}
            else:
                t_rccl = max_over_ranks(t_rccl)
                p_torch, _ = step(k - 1)
                extra["rccl_route"] = {"comm": "RcclComm (svmc_rccl_* through the C ABI)", "value": n_total * nb / t_rccl,
                                       "ms_per_step": 1e3 * t_rccl, "steps": k, "origin": rc.origin(),
                                       "prices_equal_torch_route": bool(all(np.array_equal(a, b) for a, b in zip(p_rccl, p_torch)))}
                if not args.no_c_abi_route:
                    # ... and through the fused C driver with that communicator attached (svmc_session_set_comm): what
                    # examples/price_chain_rccl.c does, one C-ABI call per chain and rank
                    try:
                        leg = c_abi_route_leg(lambda: CAbiChain(svlib, wl, P, n_local, rc.handle, rank, world, n_total, offset),
                                              step, k, n_total, nb, barrier, max_over_ranks)
                    except Exception as exc:                 # noqa: BLE001
                        leg, err = None, f"{type(exc).__name__}: {exc}"[:300]
                    extra["c_abi_route"] = leg if all_ranks_ok(leg is not None) else {"error": err or "failed on another rank"}
            rc.close()
    if isinstance(comm, svdist.RcclComm) and extra["rccl_ranks_seen"] != world:
        failures.append(f"rccl_ranks_seen {extra['rccl_ranks_seen']} != --gpus {world} on the RCCL rung")
    if ladder is not None:
        extra["comm_ladder"] = {"rung": ladder["rung"], "control_plane": ladder["control_plane"],
                                "probes": [[p_["rung"], bool(p_["ok"]), p_["seconds"]] for p_ in ladder["probes"]]}
        extra["comm_fallback_reason"] = ladder["comm_fallback_reason"]

    if world > 1 and not args.no_self_check:
        # ---- the line proves itself: the sharded job against the WHOLE job on one device (rank 0's), same seed.  The paths
        # are keyed by their global id, so the two differ by the order of the reductions' additions only
        wd.arm(t_leg, "self-check: sharded job vs the whole job on one GPU")
        p_sh, e_sh = step(777)
        check = {"seed": 20240602 + 777, "tolerance": 1e-12}
        if rank == 0:
            p_one, e_one = price(sv, wl, P, n_total, 20240602 + 777, comm=single)
            check["sharded_vs_one_gpu_max_rel_dev"] = max_rel_dev(p_sh, p_one)
            check["sharded_vs_one_gpu_max_rel_dev_stderr"] = max_rel_dev(e_sh, e_one)
            worst = max(check["sharded_vs_one_gpu_max_rel_dev"], check["sharded_vs_one_gpu_max_rel_dev_stderr"])
            if not worst <= check["tolerance"]:
                failures.append(f"sharded prices deviate from the one-GPU job by {worst:.3e} > 1e-12")
        # every rank must hold the same prices (they all hold the all-reduced sums)
        digest = float(np.sum([np.sum(a) for a in p_sh]))
        spread = gather_over_ranks(digest)
        check["ranks_agree_bitwise"] = bool(all(v == spread[0] for v in spread))
        if not check["ranks_agree_bitwise"]:
            failures.append("the ranks returned different prices")
        extra["self_check"] = check
        if rank == 0:
            extra["sharded_vs_one_gpu_max_rel_dev"] = check["sharded_vs_one_gpu_max_rel_dev"]
        if grouped:
            torch.distributed.barrier()
    if not args.no_extra_legs and (world > 1 or cfg == "c4"):
        wd.arm(2 * t_leg, "n1_share / full-job-on-one-GPU legs")
        # (a) this rank's shard WITHOUT the group: same kernels, no collectives -- the N = 1 rate the weak-scaling ratio
        #     is formed from (all ranks run it concurrently, each on its own GPU; the slowest rank's figure is reported)
        k = max(3, min(args.steps, 20))
        for i in range(2):
            price(sv, wl, P, n_local, 7 + i, comm=single)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            price(sv, wl, P, n_local, 20240602 + i, comm=single)
        torch.cuda.synchronize()
        t_share = max_over_ranks((time.perf_counter() - t0) / k)
        extra["n1_share_value"] = n_local * nb / t_share
        extra["n1_share_ms_per_step"] = 1e3 * t_share
        extra["weak_scaling_ratio"] = value / (world * extra["n1_share_value"])
        # (b) the FULL job on ONE device (rank 0's): the strong-scaling denominator
        if rank == 0:
            k = max(2, min(args.steps, 5))
            for i in range(2):
                price(sv, wl, P, n_total, 7 + i, comm=single)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                price(sv, wl, P, n_total, 20240602 + i, comm=single)
            torch.cuda.synchronize()
            t_full = (time.perf_counter() - t0) / k
            extra["c4_full_one_gpu" if cfg == "c4" else "full_job_one_gpu"] = {
                "paths": n_total, "value": n_total * nb / t_full, "ms_per_step": 1e3 * t_full,
                "speedup_of_this_run": value / (n_total * nb / t_full)}
        if grouped:
            torch.distributed.barrier()

    if world > 1 and not args.no_single_process_leg and not args.no_extra_legs:
        # ---- the same job by ONE process driving all the devices (svmc_multi_*), as a child of rank 0; the other ranks wait
        wd.arm(3 * t_leg + t_init, "single_process_route (a child process of rank 0)")
        leg = None
        if rank == 0:
            k = max(3, min(args.steps, 10))
            leg = single_process_leg(args, cfg, per_gpu, world, dev_count.value, k, 2 * t_leg + t_init)
        barrier()                                      # the other ranks idle meanwhile: their devices belong to the child
        p_ref, _ = step(777)                           # the launched ranks' prices for the child's comparison seed
        if rank == 0:
            if "error" in leg:
                extra["single_process_route"] = leg
            else:
                dev = max_rel_dev([np.asarray(leg["prices_seed_777"])], [np.concatenate([np.ravel(a) for a in p_ref])])
                extra["single_process_route"] = {
                    "entry_point": "svmc_multi_logsv_chain_price (one process, a host thread and a session per device)",
                    "value": leg["value"], "ms_per_step": leg["ms_per_step"], "steps": leg["steps"], "reduce": leg["reduce"],
                    "reduce_fallback_reason": leg.get("reduce_fallback_reason"), "rccl_ranks_seen": leg.get("rccl_ranks_seen"),
                    "devices": leg["devices"], "shards_agree_bitwise": leg["shards_agree_bitwise"],
                    "shard_ms_last_call": leg["shard_ms_last_call"], "max_rel_dev_vs_launched_ranks": dev,
                    "speedup_over_the_launched_ranks": leg["value"] / value}
                if not (dev <= 1e-12 and leg["shards_agree_bitwise"]):
                    failures.append(f"single-process route deviates from the launched ranks by {dev:.3e} (or its shards disagree)")
        if grouped:
            torch.distributed.barrier()

    result = None
    if rank == 0:
        wd.arm(2 * t_leg, "roofline legs + CPU baselines (rank 0)")
        k_ms = float(np.mean(kernel_ms))
        result = {
            "metric": "MC path-steps/sec", "value": value, "unit": "path-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["label"], "paths_per_gpu": per_gpu, "paths_total": n_total, "time_steps": nb,
                       "expiries": len(wl["grids"]), "strikes": wl["n_strikes"], "parallelism": f"path-sharded x{world}"},
            "option_prices_per_s": wl["n_strikes"] * args.steps / elapsed,
        }
        clock_mhz = None
        if world == 1 and not args.no_extra_legs:
            # the engine clock the workload sustains: the SMU's reading lags by about a second and reading it costs the
            # driver milliseconds, so it is sampled in a leg of its OWN -- the same call repeated for ~2 s after the timed
            # region -- never inside it
            with ClockPoller(int(os.environ.get("LOCAL_RANK", "0"))) as clk:
                t_end = time.perf_counter() + 2.0
                i = 0
                while time.perf_counter() < t_end:
                    step(10_000 + i)
                    i += 1
            clock_mhz = clk.steady_mhz()
        result.update(kernel_rooflines(kernel, k_ms, len(kernel_ms), n_local, wl, isa, pmc, clock_mhz, list(stamps)))
        result.update(extra)
        result["rng_stream_version"] = int(svlib.load().svmc_rng_stream_version())
        result["device_prewarm_steps"] = PREWARM
        result["gc_frozen_before_timed_region"] = True
        result["self_launched"] = os.environ.get("SVMC_BENCH_SELF_LAUNCHED") == "1"
        result["ms_per_step_profile"] = {"first5": [round(float(v), 3) for v in per_step_ms[:5]],
                                         "last5": [round(float(v), 3) for v in per_step_ms[-5:]],
                                         "median": round(float(np.median(per_step_ms)), 3),
                                         "final_barrier_ms": round(final_barrier_ms, 3),
                                         "slowest": [[int(i), round(float(per_step_ms[i]), 3)]
                                                     for i in np.argsort(per_step_ms)[::-1][:4]]}
        result["prices_head"] = [float(v) for v in prices[0][:3]]
        result["stderr_head"] = [float(v) for v in stderrs[0][:3]]
        if world == 1 and cfg == "c2" and not args.no_streamed:
            result["roofline_streamed"] = streamed_roofline(eng, P, 1024, pmc)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample_paths, P)
            if world == 1 and not args.no_extra_legs:
                result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(1024, P)
                result["cpu_baseline_numpy"] = cpu_baseline_numpy(1024, P)
        if failures:
            result["self_check_failed"] = failures
        if world == 1 and cfg == "c2" and not args.no_secondary and not args.no_extra_legs:
            # LAST key of the line (the part a truncating log keeps): the other BASELINE configurations, compact
            wd.arm(2 * t_leg, "secondary block (C1, C3, C5, f3)")
            try:
                result["secondary"] = secondary_block(sv, P)
            except Exception as exc:                         # noqa: BLE001  -- a secondary leg never costs the headline line
                result["secondary"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    # ranks > 0 get here while rank 0 still runs its roofline legs and the CPU baselines: their deadline covers that
    wd.arm(t_init if rank == 0 else 2 * t_leg + t_init, "final barrier + process-group teardown")
    n_fail = len(failures)
    if grouped:                                # world > 1, or a lone rank under SVMC_DIST_SINGLE_RANK_GROUP=1
        n_fail = int(round(max_over_ranks(float(n_fail))))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    wd.disarm()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if n_fail:
        for f in failures:
            sys.stderr.write(f"bench.py rank {rank}: SELF-CHECK FAILED: {f}\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
