#!/usr/bin/env python
"""
bench.py -- MC path-steps/s of the StochVolModels Monte Carlo hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): LogSV quadratic-drift MC, LOGSV_BTC_PARAMS, 2^20 paths per GPU x
1024 log-Euler steps (ttm = 1, 1023 steps/yr -> int(1023)+1 = 1024, dt = 2^-10), one expiry x 21 strikes
(linspace(0.5, 1.5, 21), puts below the forward, calls at/above), spot measure, LOG_RETURN, on-device Philox
randoms.  One "step" = one complete logsv_mc_chain_pricer call: state init, stepping kernel, spot-sum and
payoff reductions, (N>1: the two all-reduces over RCCL), D2H of the 21 prices.  Weak scaling: per-GPU work is
fixed, so N GPUs price N * 2^20 paths (one path set, sharded by global path id).

Prints ONE JSON line (rank 0): value = whole-job path-steps/s, plus `roofline` (dominant kernel, HIP events
on the launch stream) and `cpu_baseline` (the CPU oracle = a port of the reference's algorithm, timed here on
one host core on a bounded sample).  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before HIP initialises

import numpy as np  # noqa: E402

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8 TB/s spec; vector fp64 78.6 TFLOP/s (FMA = 2)
HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TFLOPS = 78.6
PREWARM = 10                    # untimed steps ahead of the caller's warm-up: the GPU clock ramp (see main)
# algorithmic work per LogSV path-step in fp64 op-equivalents (34 simple flops + div 10 +
# sqrt 10 + exp/log/sincos 25 each + Philox/conversion ~ 11)
LOGSV_FLOP_EQ_PER_PATH_STEP = 140.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--paths-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--nb-steps", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true")
    ap.add_argument("--cpu-sample-paths", type=int, default=1 << 19)
    return ap.parse_args()


def cpu_baseline(nb_steps: int, n_sample: int, params, strikes, types) -> dict:
    """the reference's algorithm on ONE host core (the reference is single-threaded): materialise W0, W1
    [nb_steps, n] with MT19937 + polar normals (NumPy's legacy RandomState == the reference's generator family),
    then the step-major fp64 loop and the payoff pass -- oracle/svmc_oracle.c (kind "port").  Done in chunks of
    2^17 paths (2.1 GB of normals each), like BASELINE.md section 3 prescribes."""
    from oracle import oracle
    oracle.build()
    dt = 1.0 / nb_steps
    chunk = min(n_sample, 1 << 17)
    t_rng = t_all = 0.0
    rng = np.random.RandomState(10)
    done = 0
    pooled = []
    while done < n_sample:
        t0 = time.perf_counter()
        W0 = rng.normal(0, 1, size=(nb_steps, chunk))
        W1 = rng.normal(0, 1, size=(nb_steps, chunk))
        t1 = time.perf_counter()
        x, s, q = oracle.logsv_terminal_w(np.zeros(chunk), params.sigma0 * np.ones(chunk), np.zeros(chunk), dt,
                                          params.theta, params.kappa1, params.kappa2, params.beta, params.volvol,
                                          W0, W1)
        pr, sd = oracle.payoff(x, q, 1.0, 1.0, strikes, types)
        t2 = time.perf_counter()
        pooled.append(pr)
        t_rng += t1 - t0
        t_all += t2 - t0
        done += chunk
        del W0, W1
    return {"value": done * nb_steps / t_all, "unit": "path-steps/s", "cores": 1, "kind": "port",
            "sample": f"{done} paths x {nb_steps} steps in chunks of {chunk}, 21 strikes; RandomState normals "
                      f"{t_rng:.1f}s of {t_all:.1f}s; host cores available: {os.cpu_count()}",
            # mean over the chunks; NB the estimator recentres by the SAMPLE mean of S_T, which is heavy-tailed under the
            # BTC parameters, so chunk prices scatter by more than the reported stderr (bit-level parity with the GPU
            # is established on identical randoms in tests/, not from these independent samples)
            "prices_head": [float(v) for v in np.mean(pooled, axis=0)[:3]],
            "prices_head_chunk_scatter": [float(v) for v in np.std(pooled, axis=0)[:3]]}


def cpu_baseline_all_cores(nb_steps: int, params) -> dict:
    """the same fp64 step on ALL host cores with the counter-based (Philox + libm Box-Muller) draw generated on
    the fly, OpenMP over paths (oracle svo_logsv_terminal_rng) -- not the reference's algorithm (which is
    serial), but the fairest CPU number for the GPU kernel's own algorithm."""
    from oracle import oracle
    oracle.build()
    cores = oracle.set_threads(oracle.effective_cores())      # affinity capped by the cgroup quota
    n = min(1 << 22, max(1 << 14, (cores * (1 << 16))))
    x0, s0, q0 = np.zeros(n), params.sigma0 * np.ones(n), np.zeros(n)
    t0 = time.perf_counter()
    oracle.logsv_terminal_rng(x0, s0, q0, nb_steps, 1.0 / nb_steps, params.theta, params.kappa1, params.kappa2,
                              params.beta, params.volvol, 7)
    t = time.perf_counter() - t0
    return {"value": n * nb_steps / t, "unit": "path-steps/s", "cores": cores, "kind": "port (counter-based draw, OpenMP)",
            "sample": f"{n} paths x {nb_steps} steps, stepping only, {t:.1f}s; os.cpu_count() = {os.cpu_count()}"}


def pmc_traffic(kernel: str, n_paths: int, nb_steps: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json; FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command and corrected as the microarch guide
    prescribes).  Only returned when the profile was taken at the configuration being benchmarked."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            prof = json.load(fh)
        if prof["config"] == {"paths": n_paths, "steps": nb_steps}:
            return float(prof[kernel]["hbm_bytes"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def streamed_roofline(eng, params, nb_steps: int) -> dict:
    """the fixed-randoms kernel (logsv_w_kernel) reads 16 B per path-step from HBM: the HBM-bound leg."""
    n = eng.n_path
    w0, w1 = eng.fill_normals(nb_steps, 99)
    eng.fill_state(0.0, params.sigma0, 0.0)
    dt = 1.0 / 1024
    for _ in range(2):
        eng.logsv_w(nb_steps, dt, params.theta, params.kappa1, params.kappa2, params.beta, params.volvol, 1.0, True,
                    w0, w1)
    eng.synchronize()
    eng.start_kernel_timing()
    for _ in range(5):
        eng.fill_state(0.0, params.sigma0, 0.0)
        eng.logsv_w(nb_steps, dt, params.theta, params.kappa1, params.kappa2, params.beta, params.volvol, 1.0, True,
                    w0, w1)
    ms = float(np.mean(eng.stop_kernel_timing()["logsv_w_kernel"]))
    alg_bytes = (16.0 * nb_steps + 48.0) * n
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"kernel": "logsv_w_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("logsv_w_kernel", n, nb_steps),
            "algorithmic_bytes": alg_bytes, "ms_per_launch": ms, "path_steps_per_s": n * nb_steps / (ms * 1e-3),
            "config": {"paths": n, "steps": nb_steps, "bytes_per_path_step": 16}}


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

    P = sv.LOGSV_BTC_PARAMS
    n_total = args.paths_per_gpu * world
    spy = args.nb_steps - 1                       # ttm = 1.0: int(1.0 * spy) + 1 = nb_steps
    strikes = np.linspace(0.5, 1.5, 21)
    types = np.where(strikes >= 1.0, "C", "P")
    chain = sv.OptionChain.slice_to_chain(ttm=1.0, forward=1.0, strikes=strikes, optiontypes=types)
    pricer = sv.LogSVPricer()
    nb, dt, _ = sv.set_time_grid(1.0, spy)
    assert nb == args.nb_steps

    def step(i):
        return pricer.model_mc_price_chain(chain, P, nb_path=n_total, nb_steps=spy, seed=20240602 + i)

    # The first launches after the process starts run 20-25 % slow while the GPU's clocks come up (rocprofv3: 3.8 ms
    # for the first two, the steady 3.0-3.1 ms from about the tenth).  PREWARM untimed steps bring the device to its
    # steady state before the W warm-up steps the caller asked for, so that a small W does not time the ramp.
    for i in range(PREWARM):
        step(-1000 - i)
    for i in range(args.warmup):
        step(-1 - i)
    offset, n_local = svdist.shard_range(n_total, comm.rank, comm.world)
    eng = get_engine(n_local, path_offset=offset)
    barrier()
    eng.start_kernel_timing()
    t0 = time.perf_counter()
    for i in range(args.steps):
        prices, stderrs = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = eng.stop_kernel_timing().get("logsv_rng_kernel", [float("nan")])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    path_steps = float(n_total) * nb * args.steps
    value = path_steps / elapsed
    result = None
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        # dominant kernel: logsv_rng_kernel.  Algorithmic HBM bytes per launch: state read +
        # terminal write (+ the fused snapshot), nothing inside the time loop.
        alg_bytes = 56.0 * n_local    # 24 B state read + 24 B state write + 8 B terminal-x snapshot (fused epilogue)
        hbm_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        kernel_rate = n_local * nb / (k_ms * 1e-3)
        valu_tflops = kernel_rate * LOGSV_FLOP_EQ_PER_PATH_STEP / 1e12
        result = {
            "metric": "MC path-steps/sec", "value": value, "unit": "path-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2 LogSV quadratic-drift MC, LOGSV_BTC_PARAMS, on-device Philox randoms",
                       "paths_per_gpu": args.paths_per_gpu, "paths_total": n_total, "time_steps": nb,
                       "expiries": 1, "strikes": 21, "parallelism": f"path-sharded x{world}"},
            "option_prices_per_s": 21 * args.steps / elapsed,
            "roofline": {"kernel": "logsv_rng_kernel", "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("logsv_rng_kernel", n_local, nb), "algorithmic_bytes": alg_bytes,
                         "ms_per_launch": k_ms, "launches": len(kernel_ms),
                         "note": "on-device-RNG stepping moves only 56 B per path per expiry: not HBM-bound by "
                                 "construction; the binding roof is fp64 VALU (roofline_valu)"},
            "roofline_valu": {"kernel": "logsv_rng_kernel", "bound": "valu_fp64", "achieved": valu_tflops,
                              "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": valu_tflops / FP64_VALU_PEAK_TFLOPS,
                              "flop_eq_per_path_step": LOGSV_FLOP_EQ_PER_PATH_STEP,
                              "kernel_path_steps_per_s": kernel_rate},
            "device_prewarm_steps": PREWARM,
            "prices_head": [float(v) for v in prices[0][:3]],
            "stderr_head": [float(v) for v in stderrs[0][:3]],
        }
        if world == 1 and not args.no_streamed:
            result["roofline_streamed"] = streamed_roofline(eng, P, min(args.nb_steps, 1024))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(nb, args.cpu_sample_paths, P, strikes, types)
            result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(nb, P)
    if torch.distributed.is_initialized():     # world > 1, or a lone rank under SVMC_DIST_SINGLE_RANK_GROUP=1
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
