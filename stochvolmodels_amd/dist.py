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
