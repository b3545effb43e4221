#!/usr/bin/env python
"""
bench.py -- MC path-steps/s of the StochVolModels Monte Carlo hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

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
that binds it (VALU issue: the kernel moves 56 B per path per expiry and no HBM byte inside the time loop), HIP events
on the launch stream; `roofline_hbm` = the same kernel against HBM (BASELINE asks for it); `cpu_baseline` = the CPU
oracle (a port of the reference's algorithm) timed on one host core on a bounded sample; at N > 1 also
`n1_share_value` (this rank's shard priced WITHOUT the group: the denominator of the weak-scaling ratio) and
`c4_full_one_gpu` (all N x 2^21 paths on ONE device: the denominator of the strong-scaling ratio).
See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before HIP initialises

import numpy as np  # noqa: E402

# MI355X (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8 TB/s spec; 256 CUs x 4 SIMDs, max clock 2400 MHz, one VALU
# instruction per SIMD per 4 cycles (wave64 on 16 lanes); vector fp64 78.6 TFLOP/s (FMA = 2)
HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TFLOPS = 78.6
N_SIMD = 256 * 4
MAX_CLOCK_HZ = 2.4e9
VALU_ISSUE_PEAK = N_SIMD * MAX_CLOCK_HZ / 4.0          # full-rate wave-instructions per second, whole chip
# Untimed steps ahead of the caller's warm-up (disclosed as device_prewarm_steps).  Two things have to be behind the
# process before K steps of 2-6 ms can be timed: the GPU's clock ramp (the first ~10 launches run 20-25 % slow) and a
# ONE-TIME stall of the ROCm runtime -- a single call of ~60 ms observed at the ~150th chain call of a process for C2
# and the ~41st for C4 (profiles/r02_runtime_stall.txt: 400-step runs, one slow step each, none afterwards) -- which a
# 50-step window would otherwise swallow whole.  `ms_per_step_profile` in the output shows the timed steps one by one.
PREWARM = int(os.environ.get("SVMC_BENCH_PREWARM", "200"))       # the environment override exists for the test suite
# SURVEY.md 8(d)'s ESTIMATE of the algorithmic work per LogSV path-step in fp64 op-equivalents (34 simple flops + div 10
# + sqrt 10 + exp/log/sincos 25 each + Philox/conversion ~ 11): a model of the reference's arithmetic, not a count
LOGSV_FLOP_EQ_PER_PATH_STEP = 140.0


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
    ap.add_argument("--cpu-sample-paths", type=int, default=1 << 19)
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


def price(sv, wl, P, n_path, seed, comm=None):
    return sv.logsv_mc_chain_pricer(ttms=wl["ttms"], forwards=wl["forwards"], discfactors=wl["dfs"],
                                    strikes_ttms=wl["strikes"], optiontypes_ttms=wl["types"], v0=P.sigma0, theta=P.theta,
                                    kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta, volvol=P.volvol,
                                    vol_backbone_etas=np.ones(len(wl["ttms"])), nb_path=n_path,
                                    nb_steps_per_year=wl["spy"], seed=seed, comm=comm)


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
def load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "r02_pmc.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


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


def kernel_rooflines(kernel: str, k_ms: float, launches: int, n_local: int, wl, pmc, clock_mhz) -> dict:
    """the stepping kernel of one chain call against (a) the VALU issue port -- the roof that binds it -- from the
    committed SQ_INSTS_VALU pass, (b) HBM, (c) SURVEY's flop-equivalent estimate"""
    nb, m = wl["nb_total"], len(wl["grids"])
    prof = pmc.get(kernel, {})
    per_step = prof.get("valu_insts_per_wave_step")
    quarter = prof.get("quarter_rate_insts_per_step", 2)
    wave_steps = (n_local / 64.0) * nb
    out = {}
    if per_step is not None:
        # full-rate-equivalent issue slots: v_rcp_f64 / v_rsq_f64 hold the port for 16 cycles instead of 4
        slots = per_step + 3.0 * quarter
        achieved = slots * wave_steps / (k_ms * 1e-3)
        out["roofline"] = {
            "kernel": kernel, "bound": "valu_issue", "achieved": achieved, "peak": VALU_ISSUE_PEAK,
            "unit": "wave-instr/s (full-rate slots)", "frac": achieved / VALU_ISSUE_PEAK,
            "insts_per_wave_step": per_step, "quarter_rate_insts": quarter, "issue_slots_per_wave_step": slots,
            "insts_source": prof.get("source", "profiles/r02_pmc.json (rocprofv3 --pmc SQ_INSTS_VALU)"),
            "peak_definition": "1024 SIMDs x 2400 MHz / 4 cycles per wave64 instruction",
            # informational: the SMU's engine-clock reading sampled while the same call repeats for 2 s after the timed region;
            # the sensor averages and lags (boxes of the pool have reported 2100-2395 MHz for the same kernel time), so the
            # fraction above is against the 2400 MHz maximum, never against this reading
            "clock_mhz_sensor": clock_mhz, "ms_per_launch": k_ms, "launches": launches,
            "traffic": prof.get("hbm_bytes") if prof.get("config") == {"paths": n_local, "steps": nb} else None,
            "algorithmic_bytes": (48.0 + 8.0 * m) * n_local,
        }
        if prof.get("note"):
            out["roofline"]["traffic_note"] = prof["note"]
    alg_bytes = (48.0 + 8.0 * m) * n_local     # 24 B state read + 24 B write per chain, 8 B terminal-x snapshot per expiry
    hbm_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
    hbm = {"kernel": kernel, "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": hbm_gbs / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes, "ms_per_launch": k_ms, "launches": launches,
           "traffic": prof.get("hbm_bytes") if prof.get("config") == {"paths": n_local, "steps": nb} else None,
           "note": "on-device-RNG stepping moves 48 + 8 M bytes per path per chain and nothing inside the time loop: "
                   "not HBM-bound by construction"}
    out["roofline_hbm"] = hbm
    if "roofline" not in out:                  # no committed counter pass for this kernel: report the HBM roof
        out["roofline"] = hbm
    rate = n_local * nb / (k_ms * 1e-3)
    out["roofline_valu_flop_estimate"] = {
        "kernel": kernel, "bound": "valu_fp64", "achieved": rate * LOGSV_FLOP_EQ_PER_PATH_STEP / 1e12,
        "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": rate * LOGSV_FLOP_EQ_PER_PATH_STEP / 1e12 / FP64_VALU_PEAK_TFLOPS,
        "flop_eq_per_path_step": LOGSV_FLOP_EQ_PER_PATH_STEP, "kernel_path_steps_per_s": rate,
        "note": "ESTIMATE: SURVEY.md 8d's model of the reference's arithmetic (140 fp64 op-equivalents per path-step), "
                "not instructions this kernel executes -- it rises as the kernel does less; `roofline` is the measured one"}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    import torch  # device plumbing + torch.distributed only
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import dist as svdist
    from stochvolmodels_amd.engine import get_engine

    comm = svdist.init_from_env()
    if world == 1:
        torch.cuda.set_device(0)

    def barrier():
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v: float) -> float:
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64,
                         device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    P = sv.LOGSV_BTC_PARAMS
    cfg = args.config or ("c2" if world == 1 else "c4")
    wl = make_workload(cfg, sv)
    per_gpu = args.paths_per_gpu or ((1 << 20) if cfg == "c2" else (1 << 21))
    n_total = per_gpu * world
    nb = wl["nb_total"]
    kernel = "logsv_rng_kernel" if len(wl["grids"]) == 1 else "logsv_chain_rng_kernel"
    pmc = load_pmc()

    def step(i):
        return price(sv, wl, P, n_total, 20240602 + i)

    for i in range(PREWARM):
        step(-1000 - i)
    for i in range(args.warmup):
        step(-1 - i)
    offset, n_local = svdist.shard_range(n_total, comm.rank, comm.world)
    eng = get_engine(n_local, path_offset=offset)
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
    elapsed = max_over_ranks(elapsed)
    value = float(n_total) * nb * args.steps / elapsed

    extra = {}
    single = svdist.SingleComm()
    if world > 1 and isinstance(comm, svdist.TorchComm):
        # the stream-ordering shortcut of TorchComm (no host synchronisation around the all-reduces) against the
        # host-synchronised fallback, with real peers: the bits must be identical
        os.environ["SVMC_DIST_STRICT_SYNC"] = "1"
        p_strict, _ = step(424242)
        os.environ["SVMC_DIST_STRICT_SYNC"] = "0"
        p_ordered, _ = step(424242)
        extra["stream_ordered_equals_strict_sync"] = bool(all(np.array_equal(a, b) for a, b in zip(p_strict, p_ordered)))
    extra["comm"] = type(comm).__name__
    if not args.no_extra_legs and (world > 1 or cfg == "c4"):
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
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    result = None
    if rank == 0:
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
        result.update(kernel_rooflines(kernel, k_ms, len(kernel_ms), n_local, wl, pmc, clock_mhz))
        result.update(extra)
        result["device_prewarm_steps"] = PREWARM
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
    if torch.distributed.is_initialized():     # world > 1, or a lone rank under SVMC_DIST_SINGLE_RANK_GROUP=1
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
