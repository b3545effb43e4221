"""
Path sharding over the GPUs of one node.

Paths are independent, so rank r of R owns the global path ids [r*N/R, (r+1)*N/R) and its own resident
state; the Philox counter is the GLOBAL path id, so the union of the shards is the same path set for any
R.  The only cross-rank couplings are the two reductions of compute_mc_vars_payoff
(utils/mc_payoffs.py:61-63 and :85-86): per chain ONE all-reduce of [sum F*exp(x), count] per expiry
(2*M doubles) and ONE all-reduce of [sum d, sum d^2, count] per strike (3*sum(K_i) doubles), packed fp64,
over RCCL (torch.distributed backend "nccl") -- a few KB, latency-bound on xGMI.  SURVEY.md 8(e).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global path id, number of local paths) of `rank`."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi - lo


class SingleComm:
    """one process, one GPU: reductions are already global."""
    rank = 0
    world = 1

    def alloc(self, engine, n_doubles: int, tag: str):
        return engine.alloc_sums(n_doubles, tag)

    def all_reduce_sum(self, engine, handle) -> None:
        return None

    def to_host(self, engine, ptr, handle, n: int):
        return engine.download(ptr, n)


class TorchComm:
    """torch.distributed process group (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

    The reduction buffers are torch tensors on the engine's device so the collective runs on them in
    place; libsvmc's kernels write them through `data_ptr()`."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def _device(self, engine):
        kind = getattr(engine, "torch_device", None)
        if kind is not None:
            return self._torch.device(kind)
        return self._torch.device("cuda", engine.device)

    def alloc(self, engine, n_doubles: int, tag: str):
        # one persistent tensor per (engine, tag), zero-initialised; the HANDLE is the live prefix t[:n_doubles], so
        # the collective reduces exactly the doubles this chain wrote -- every rank calls with the same n_doubles
        # (it is a function of the chain alone), whatever chains each rank priced before.  The tensors live ON the engine
        # object (not in a table keyed by id(engine): a closed engine's id can come back as a new engine on another device,
        # and the cached tensor would then sit on the wrong GPU), keyed by this communicator, the tag and the device
        device = self._device(engine)
        bufs = engine.__dict__.setdefault("_comm_bufs", {})
        key = (id(self), tag, str(device))
        n = max(int(n_doubles), 1)
        t = bufs.get(key)
        if t is None or t.numel() < n:
            t = self._torch.zeros(n, dtype=self._torch.float64, device=device)
            bufs[key] = t
        return t.data_ptr(), t.narrow(0, 0, n)

    def _stream_ordered(self, engine, handle) -> bool:
        """True when libsvmc's launches and the collective are ordered by the stream alone: the engine launches on
        the very stream torch regards as current on that device (the default stream, unless someone changed either).
        torch.distributed then does what it does for torch's own kernels -- the collective waits for the prior work
        of the current stream and the current stream waits for the collective -- and no host synchronisation is
        needed.  SVMC_DIST_STRICT_SYNC=1 forces the host synchronisations regardless."""
        if not handle.is_cuda or os.environ.get("SVMC_DIST_STRICT_SYNC") == "1":
            return False
        mine = getattr(engine, "stream", None)
        mine = int(getattr(mine, "value", mine) or 0)
        return int(self._torch.cuda.current_stream(handle.device).cuda_stream) == mine

    def all_reduce_sum(self, engine, handle) -> None:
        handle = handle.view(-1)
        ordered = self._stream_ordered(engine, handle)
        if not ordered:
            engine.synchronize()                   # svmc kernels that wrote the buffer are complete
        self._dist.all_reduce(handle, op=self._dist.ReduceOp.SUM, group=self.group)
        if handle.is_cuda and not ordered:
            self._torch.cuda.synchronize(handle.device)   # reduced values visible to the next svmc kernels

    def to_host(self, engine, ptr, handle, n: int):
        if handle.is_cuda and self._stream_ordered(engine, handle) and hasattr(engine, "download"):
            # the reduced sums sit in device memory behind `ptr` and the collective is ordered on the engine's stream: take them
            # through the engine's page-locked landing buffer like the single-GPU route does (a torch .cpu() of 4 KB goes through
            # the runtime's pageable staging path: ~16 us more per chain, tools/ubench/sync_latency.py)
            return engine.download(ptr, n)
        engine.synchronize()
        return handle[:n].detach().cpu().numpy().copy()


class RcclComm:
    """RCCL through libsvmc's own C ABI (include/svmc.h svmc_rccl_*): the reduction buffers are plain libsvmc device
    allocations and the collectives are issued by libsvmc on the engine's stream, stream-ordered against its kernels
    -- no torch tensor and no torch.distributed call in the data path.  The second route beside TorchComm; the one a
    C / C++ host takes (examples/price_chain_rccl.c) driven from Python.

    `id_bytes`: the SVMC_RCCL_UNIQUE_ID_BYTES of rank 0's svmc_rccl_unique_id(), identical on all ranks (ship them
    by any means; init_from_env(comm="rccl") broadcasts them over a gloo group)."""

    def __init__(self, rank: int, world: int, id_bytes: bytes):
        import ctypes as C

        from . import _lib
        self._C, self._lib = C, _lib
        self.rank, self.world = int(rank), int(world)
        L = _lib.load()
        if not L.svmc_rccl_available():
            raise _lib.SvmcError("RCCL unavailable: " + L.svmc_rccl_origin().decode())
        buf = C.create_string_buffer(bytes(id_bytes), len(id_bytes))
        comm = C.c_void_p()
        _lib.check(L.svmc_rccl_comm_create(C.byref(comm), buf, len(id_bytes), self.world, self.rank))
        self.handle = comm
        # RCCL builds its rings lazily at the first collective: do that here, not inside the first chain priced
        warm = C.c_void_p()
        _lib.check(L.svmc_malloc(C.byref(warm), 8))
        _lib.check(L.svmc_memset(warm, 0, 8, None))
        _lib.check(L.svmc_rccl_all_reduce_sum(self.handle, warm, 1, None))
        _lib.check(L.svmc_stream_synchronize(None))
        _lib.check(L.svmc_free(warm))

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().svmc_rccl_unique_id(buf, 128))
        return bytes(buf.raw)

    def alloc(self, engine, n_doubles: int, tag: str):
        ptr, _owner = engine.alloc_sums(max(int(n_doubles), 1), "rccl_" + tag)
        return ptr, (ptr, int(n_doubles))

    def all_reduce_sum(self, engine, handle) -> None:
        ptr, n = handle
        self._lib.check(self._lib.load().svmc_rccl_all_reduce_sum(self.handle, ptr, n, engine.stream))

    def to_host(self, engine, ptr, handle, n: int):
        return engine.download(ptr, n)

    def ranks_seen(self) -> int:
        """ncclCommCount of the communicator: the number of ranks RCCL itself sees"""
        n = self._C.c_int(0)
        self._lib.check(self._lib.load().svmc_rccl_comm_count(self.handle, self._C.byref(n), None))
        return int(n.value)

    def origin(self) -> str:
        return self._lib.load().svmc_rccl_origin().decode()

    def close(self) -> None:
        if self.handle is not None:
            self._lib.load().svmc_rccl_comm_destroy(self.handle)
            self.handle = None


_default_comm = SingleComm()


def get_default_comm():
    return _default_comm


def set_default_comm(comm) -> None:
    global _default_comm
    _default_comm = comm if comm is not None else SingleComm()


def init_from_env(backend: Optional[str] = None, comm: Optional[str] = None, on_phase=None):
    """one process per GPU launched by torch.distributed.run: bind LOCAL_RANK's GPU, create the process
    group over RCCL and make it the default communicator of the chain pricers.

    comm = "torch" (default): the collectives go through torch.distributed (backend "nccl" = RCCL);
    comm = "rccl" (or SVMC_DIST_COMM=rccl): they go through libsvmc's own RCCL entry points (RcclComm); the torch
    process group is then only the bootstrap (a gloo group is enough) that ships rank 0's unique id.

    on_phase(name): called at "rendezvous" (before init_process_group: waits for every rank to arrive), "collective_init"
    (before the first all-reduce: RCCL builds its communicator there) and "ready" -- a launcher's watchdog arms its
    per-phase deadlines from it (bench.py: a rank that never arrives or an RCCL init that never returns ends the run
    with the rank's traceback instead of hanging)."""
    phase = on_phase if on_phase is not None else (lambda name: None)
    comm = comm or os.environ.get("SVMC_DIST_COMM") or "torch"
    if comm == "rccl" and backend is None and os.environ.get("SVMC_DIST_BACKEND") is None:
        backend = "gloo"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("SVMC_DIST_SINGLE_RANK_GROUP") != "1":
        set_default_comm(None)
        return get_default_comm()
    # SVMC_DIST_SINGLE_RANK_GROUP=1 builds the process group for a lone rank too, so that a 1-GPU box runs the very
    # code of the N>1 path (RCCL communicator, torch-owned reduction buffers, stream-ordered collectives)
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("RANK", "0")
    # the host driver of these boxes only supports dmabuf IPC; RCCL's cross-process handles need this before HIP starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if backend is None:
        # SVMC_DIST_BACKEND=gloo lets several ranks share one GPU (tests on a 1-GPU box); RCCL is the default
        backend = os.environ.get("SVMC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        device = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(device)
        # ... and through libsvmc itself: its engines take the HIP runtime's current device, which must be this rank's GPU
        # whatever torch does about initialising lazily
        from . import _lib
        _lib.check(_lib.load().svmc_set_device(device))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        phase("rendezvous")
        if backend == "nccl":
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
        # RCCL builds its communicator (rings over xGMI) lazily at the first collective: do that here, once, not
        # inside the first chain that is priced
        phase("collective_init")
        warm = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", device) if backend == "nccl" else "cpu")
        dist.all_reduce(warm)
        if warm.is_cuda:
            torch.cuda.synchronize(warm.device)
    tcomm = TorchComm()
    _share_rng_seed(tcomm)
    if comm == "rccl":
        # bind this rank's GPU through libsvmc itself (torch may be a CPU-only build, or the bootstrap group gloo): RCCL
        # needs one device per rank
        import ctypes as C

        from . import _lib
        count = C.c_int(0)
        _lib.check(_lib.load().svmc_device_count(C.byref(count)))
        if tcomm.world > count.value and os.environ.get("SVMC_DIST_BACKEND") != "gloo":
            raise _lib.SvmcError(f"comm='rccl': {tcomm.world} ranks but {count.value} GPU(s) visible -- RCCL needs one "
                                 f"device per rank")
        _lib.check(_lib.load().svmc_set_device(local_rank % max(count.value, 1)))
        phase("collective_init")
        box = [RcclComm.unique_id() if tcomm.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        set_default_comm(RcclComm(tcomm.rank, tcomm.world, box[0]))
    else:
        set_default_comm(tcomm)
    phase("ready")
    return get_default_comm()


def _share_rng_seed(comm) -> None:
    """every rank must key Philox with the SAME (seed, call counter): the union of the shards is one path set only
    then.  An un-seeded process draws its seed from OS entropy (utils/funcs.py), so rank 0's pair is broadcast when
    the group is built; afterwards set_seed(value) / seed= must be called identically on all ranks (as any SPMD
    program does)."""
    from .utils import funcs
    t = comm._torch
    dev = "cuda" if comm._dist.get_backend(comm.group) == "nccl" else "cpu"
    seed, calls = funcs.get_rng_state()
    buf = t.tensor([seed & 0xFFFFFFFF, seed >> 32, calls], dtype=t.int64, device=dev)
    comm._dist.broadcast(buf, src=0, group=comm.group)
    lo, hi, calls = (int(v) for v in buf.cpu().tolist())
    funcs.set_rng_state((hi << 32) | lo, calls)


# ---------------------------------------------------------------------------------------------------------------------
# A start that cannot come back empty: the collective layer chosen by a ladder of rungs, each PROBED in a child process first.
#
# The first contact of this code with an 8-GPU node must produce prices (and bench.py a number) whatever the node's RCCL does:
# an RCCL initialisation can fail (an exception: the next rung takes over) or HANG inside the runtime, where no Python timer
# can interrupt it and a watchdog can only end the rank.  So every rank first forms a gloo group over TCP on 127.0.0.1 -- the
# control plane: agreement votes, the RCCL unique id, the seed -- and then, rung by rung,
#     "nccl"  torch.distributed's RCCL backend                      (TorchComm on a second, nccl process group)
#     "rccl"  libsvmc's own RCCL entry points, include/svmc.h        (RcclComm: ncclCommInitRank below the C ABI)
#     "gloo"  the control plane itself carries the two all-reduces   (TorchComm on the gloo group: a few KB through the host)
# starts a CHILD process per rank that initialises that rung and runs one sum all-reduce under a deadline.  Only a rung whose
# probe succeeded on EVERY rank is initialised in the ranks themselves; a child that hangs is killed by its parent (its exact
# pid) and costs the deadline, nothing else.  The collectives of a chain are 4 KB: on the gloo rung the job is a few per cent
# slower and says so (`report["rung"]`, `report["comm_fallback_reason"]`).
# ---------------------------------------------------------------------------------------------------------------------
def _probe_main(rung: str) -> int:
    """child process of init_with_fallback (python -m stochvolmodels_amd.dist --probe <rung>): bring the rung up among the
    probe children of all ranks (their own rendezvous port), one sum all-reduce, exit 0 when it counted every rank"""
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    fault = os.environ.get("SVMC_BENCH_FAULT", "")
    if fault in (f"{rung}_init", f"{rung}_init:{rank}"):
        raise RuntimeError(f"fault injection: {rung} initialisation fails")
    if fault in (f"{rung}_hang", f"{rung}_hang:{rank}"):
        import time
        time.sleep(3600)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if rung == "nccl":
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to torch")
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", device_id=device)
        t = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize(device)
        seen = int(round(float(t.item())))
    elif rung == "rccl":
        import ctypes as C

        from . import _lib
        count = C.c_int(0)
        _lib.check(_lib.load().svmc_device_count(C.byref(count)))
        if count.value < 1:
            raise RuntimeError("no HIP device visible")
        _lib.check(_lib.load().svmc_set_device(local_rank % count.value))
        dist.init_process_group(backend="gloo")
        box = [RcclComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rc = RcclComm(rank, world, box[0])          # creates the communicator and runs one all-reduce
        seen = rc.ranks_seen()
        rc.close()
    else:
        raise ValueError(f"unknown rung {rung}")
    if seen != world:
        raise RuntimeError(f"{rung}: the collective saw {seen} of {world} ranks")
    dist.destroy_process_group()
    print(f"probe {rung}: ok, {seen} ranks", flush=True)
    return 0


def _run_probe(rung: str, port: int, timeout: float):
    """-> (ok, reason) of this rank's probe child for `rung`"""
    import subprocess
    import sys
    # the children rendezvous among THEMSELVES on `port`: under torch.distributed.run the ranks carry TORCHELASTIC_USE_AGENT_STORE
    # (= "connect to the launcher's store at MASTER_PORT as a client"), which on a port of our own nobody serves -- every probe
    # would wait for a store that never comes and the ladder would fall to gloo on a perfectly healthy RCCL.  Rank 0's child
    # must host the store: the launcher's variables stay with the ranks.
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        proc = subprocess.Popen([sys.executable, "-m", "stochvolmodels_amd.dist", "--probe", rung], env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    except OSError as exc:
        return False, f"probe did not start: {exc}"
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()                                   # this child, by its pid
        proc.communicate()
        return False, f"probe did not finish within {timeout:.0f} s (killed)"
    if proc.returncode == 0:
        return True, ""
    tail = [ln.strip() for ln in (out or "").strip().splitlines() if ln.strip()]
    # the child's own last words: the last line that names an error, else its last line
    import re
    exc = [ln for ln in tail if re.search(r"\b\w*(Error|Exception)\b: \S", ln)]          # "RuntimeError: ...", "DistBackendError: ..."
    said = exc or [ln for ln in tail if any(w in ln for w in ("invalid", "failed", "fault injection")) and "Warning" not in ln]
    return False, (said[-1] if said else (tail[-1] if tail else f"probe exited with status {proc.returncode}"))[:240]


def init_with_fallback(rungs=("nccl", "rccl", "gloo"), on_phase=None, probe_timeout: Optional[float] = None):
    """one process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment): form
    the gloo control plane, pick the first rung of `rungs` that works on every rank (see above), make its communicator the
    default of the chain pricers and return (comm, report).  report: {"rung", "comm", "backend", "control_plane",
    "comm_fallback_reason" (None when the first rung held), "probes": [{rung, ok, reason, seconds}], "ranks_share_a_device"}.
    A world of one returns the single-GPU communicator.  SVMC_DIST_RUNGS overrides `rungs` (comma-separated)."""
    import time
    phase = on_phase if on_phase is not None else (lambda name: None)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    report = {"rung": "single", "comm": "SingleComm", "backend": None, "control_plane": None, "comm_fallback_reason": None,
              "probes": [], "ranks_share_a_device": False}
    if world <= 1 and os.environ.get("SVMC_DIST_SINGLE_RANK_GROUP") != "1":
        set_default_comm(None)
        return get_default_comm(), report
    # SVMC_DIST_SINGLE_RANK_GROUP=1: a lone rank walks the ladder too -- probe children, the nccl group / the RCCL communicator,
    # the collectives of a chain -- so that a 1-GPU box exercises the code of the N > 1 start (tests/test_gpu_parity.py)
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("RANK", "0")
    world = max(world, 1)
    if os.environ.get("SVMC_DIST_RUNGS"):
        rungs = tuple(r.strip() for r in os.environ["SVMC_DIST_RUNGS"].split(",") if r.strip())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    n_dev = 0
    try:
        import ctypes as C

        from . import _lib
        count = C.c_int(0)
        _lib.check(_lib.load().svmc_device_count(C.byref(count)))
        n_dev = count.value
        if n_dev > 0:
            _lib.check(_lib.load().svmc_set_device(local_rank % n_dev))
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
    except Exception:                                   # a CPU-only box (the gloo tests): the engines are test doubles there
        n_dev = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    phase("rendezvous")
    if dist.is_initialized() and "gloo" not in str(dist.get_backend()):
        # the caller brought its own process group (nccl): it IS the collective layer; there is no control plane to vote on
        chosen = TorchComm()
        report.update(rung=str(dist.get_backend()), comm="TorchComm", backend=str(dist.get_backend()), control_plane=None)
        _share_rng_seed(chosen)
        set_default_comm(chosen)
        phase("ready")
        return chosen, report
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo")
    control = TorchComm()                                # the gloo world group
    report["control_plane"] = "gloo"
    # do the ranks of some NODE outnumber the devices that node's ranks see?  Counted per node -- LOCAL_WORLD_SIZE (torchrun sets
    # it; a bare launch of one node: the world) against the devices visible to this rank -- and agreed over the control plane, so
    # that two nodes of eight (WORLD_SIZE 16, 8 devices each) are not mistaken for sharing and one rank's failed device query
    # cannot send the ranks down different sequences of collectives (round-5 advisor finding)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    facts = [None] * world
    dist.all_gather_object(facts, (local_world, n_dev))
    report["ranks_share_a_device"] = any(nd > 0 and lw > nd for lw, nd in facts)
    report["ranks_without_a_device"] = [r for r, (_, nd) in enumerate(facts) if nd == 0]

    def everyone(ok: bool) -> bool:
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def first_reason(reason: str) -> str:
        box = [None] * world
        dist.all_gather_object(box, reason)
        return next((f"rank {r}: {msg}" for r, msg in enumerate(box) if msg), "")

    timeout = float(probe_timeout if probe_timeout is not None else os.environ.get("SVMC_DIST_PROBE_TIMEOUT", "180"))
    reasons, chosen = [], None
    for rung in rungs:
        t0 = time.perf_counter()
        if rung == "gloo":
            chosen = control
            report["probes"].append({"rung": rung, "ok": True, "reason": "", "seconds": 0.0})
            break
        if rung not in ("nccl", "rccl"):
            raise ValueError(f"unknown rung {rung!r}")
        # skip or probe is decided from the facts every rank holds (gathered above): all ranks take the same branch, hence issue
        # the same collectives below
        forced = os.environ.get("SVMC_DIST_FORCE_PROBE") == "1"          # (the CPU tests of the probe machinery)
        if report["ranks_without_a_device"] and not forced:
            ok, why = False, ("no HIP device visible" if n_dev == 0
                              else f"rank(s) {report['ranks_without_a_device']} see no HIP device")
        elif report["ranks_share_a_device"] and not forced:
            ok, why = False, f"{local_world} ranks of a node share {n_dev} device(s): RCCL takes one rank per device"
        else:
            port = [None]
            if rank == 0:
                import socket
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    port[0] = sk.getsockname()[1]
            dist.broadcast_object_list(port, src=0)
            phase("collective_probe")
            ok, why = _run_probe(rung, int(port[0]), timeout)
        all_ok = everyone(ok)
        why_all = "" if all_ok else (first_reason(why) or "the probe failed on another rank")
        if all_ok:
            # the probe passed everywhere: bring the rung up in the ranks themselves
            phase("collective_init")
            comm_, err = None, ""
            try:
                if rung == "nccl":
                    device = torch.device("cuda", local_rank % torch.cuda.device_count())
                    group = dist.new_group(backend="nccl", device_id=device) if _new_group_takes_device_id(dist) \
                        else dist.new_group(backend="nccl")
                    warm = torch.ones(1, dtype=torch.float64, device=device)
                    dist.all_reduce(warm, group=group)
                    torch.cuda.synchronize(device)
                    if int(round(float(warm.item()))) != world:
                        raise RuntimeError(f"the nccl group counted {warm.item():.0f} of {world} ranks")
                    comm_ = TorchComm(group)
                else:
                    box = [RcclComm.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    comm_ = RcclComm(rank, world, box[0])
            except Exception as exc:                     # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"[:240]
            all_ok = everyone(comm_ is not None)
            if all_ok:
                chosen = comm_
            else:
                why_all = first_reason(err) or "initialisation failed on another rank"
                if isinstance(comm_, RcclComm):
                    comm_.close()
        report["probes"].append({"rung": rung, "ok": bool(all_ok), "reason": why_all, "seconds": round(time.perf_counter() - t0, 2)})
        if all_ok:
            break
        reasons.append(f"{rung}: {why_all}")
    if chosen is None:
        raise RuntimeError("no collective layer could be initialised: " + "; ".join(reasons))
    report["rung"] = report["probes"][-1]["rung"]
    report["comm"] = type(chosen).__name__
    report["backend"] = "gloo" if chosen is control else ("nccl" if isinstance(chosen, TorchComm) else "rccl (libsvmc)")
    report["comm_fallback_reason"] = "; ".join(reasons) or None
    _share_rng_seed(control)
    set_default_comm(chosen)
    phase("ready")
    return chosen, report


def _new_group_takes_device_id(dist) -> bool:
    import inspect
    try:
        return "device_id" in inspect.signature(dist.new_group).parameters
    except (TypeError, ValueError):
        return False


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 3 and sys.argv[1] == "--probe":
        sys.exit(_probe_main(sys.argv[2]))
    raise SystemExit("usage: python -m stochvolmodels_amd.dist --probe nccl|rccl")
