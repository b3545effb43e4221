"""
HipEngine: the host-side owner of one GPU's Monte Carlo state and the thin caller of libsvmc's kernels.

One engine = one HIP device + one stream + the resident per-path state (x, vol, qvar) of `n_path` local
paths, the per-expiry snapshots the payoff pass reads, and the reduction scratch.  State never leaves
HBM between expiries (the reference carries it slice to slice in NumPy arrays,
pricers/logsv_pricer.py:843-856).

The chain drivers in mc_chain.py talk to an engine only through the methods below, so the sharding /
collective logic can be exercised on CPU by a test double (tests/ only); the product always uses this
class and fails loudly when libsvmc.so or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import functools
import sys
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

LOG_RETURN, Q_VAR, SIGMA = 1, 2, 3
HESTON_EULER_FLOOR, HESTON_QE = 0, 1
_TYPE_CODES = {"C": 0, "P": 1, "IC": 2, "IP": 3}


_CODES_CACHE = {}           # (dtype, bytes) of a NumPy array of option types -> its int8 codes (read-only)


def option_type_codes(optiontypes: Sequence) -> np.ndarray:
    """'C','P','IC','IP' -> int8; anything else raises like utils/mc_payoffs.py:84.  The codes of a NumPy array are kept
    (a calibration asks for the same few arrays at every objective evaluation) and come back read-only."""
    key = None
    if isinstance(optiontypes, np.ndarray) and optiontypes.dtype.kind == "U" and optiontypes.size <= 4096:
        key = (optiontypes.dtype.str, optiontypes.shape, optiontypes.tobytes())
        hit = _CODES_CACHE.get(key)
        if hit is not None:
            return hit
        if optiontypes.ndim != 1:
            out = option_type_codes(optiontypes.ravel()).reshape(optiontypes.shape)
            out.flags.writeable = False
            _CODES_CACHE[key] = out
            return out
    out = np.empty(len(optiontypes), dtype=np.int8)
    for i, t in enumerate(optiontypes):
        code = _TYPE_CODES.get(str(t))
        if code is None:
            raise ValueError("unknown option payoff code")
        out[i] = code
    if key is not None:
        if len(_CODES_CACHE) > 256:
            _CODES_CACHE.clear()
        out.flags.writeable = False
        _CODES_CACHE[key] = out
    return out


def payoff_shifts(strikes: np.ndarray, codes: np.ndarray, forward: float, variable_type: int) -> np.ndarray:
    """per-strike constants subtracted inside the payoff sums (include/svmc.h): the intrinsic value at the
    forward for LOG_RETURN (the recentred spots average to the forward exactly), zero otherwise."""
    shifts = np.zeros(len(strikes), dtype=np.float64)
    if variable_type == LOG_RETURN:
        call = (codes == 0) | (codes == 2)
        intrinsic = np.where(call, np.maximum(forward - strikes, 0.0), np.maximum(strikes - forward, 0.0))
        shifts = np.where(codes >= 2, intrinsic / forward, intrinsic).astype(np.float64)
    return np.ascontiguousarray(shifts)


class DeviceBuffer:
    """a caller-owned HBM allocation of `n` doubles (svmc_malloc / svmc_free)."""

    def __init__(self, n: int):
        self.n = int(n)
        self.nbytes = 8 * self.n
        p = C.c_void_p()
        _lib.check(_lib.load().svmc_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def free(self) -> None:
        if getattr(self, "ptr", None):
            _lib.load().svmc_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def offset(self, n_doubles: int) -> int:
        return self.ptr + 8 * int(n_doubles)


# ---- bulk results: resident, or brought to the host at the PCIe rate ---------------------------------------------------
# hipMemcpy into a pageable NumPy array goes through the runtime's small staging buffers on one host thread; an 8.6 GB
# result (simulate_vol_paths at 2^20 x 1024) then takes far longer to fetch than the 2 ms it took to compute.  A bulk
# download here is a pipeline: the device-to-host copies run back to back, asynchronously, into a ring of page-locked
# buffers the process keeps, and a few host threads move each finished chunk into the destination array while the next
# chunks are in flight (NumPy releases the GIL while it copies) -- the GPU link and the host's memory system both stay busy.
import os as _os
PIPELINE_CHUNK_BYTES = int(_os.environ.get("SVMC_PIPELINE_CHUNK_MIB", "32")) << 20
PIPELINE_SLOTS = int(_os.environ.get("SVMC_PIPELINE_SLOTS", "8"))
PIPELINE_THREADS = int(_os.environ.get("SVMC_PIPELINE_THREADS", "6"))        # swept in profiles/r05_pipeline_sweep.jsonl
PIPELINE_MIN_BYTES = 8 << 20                   # below this a plain copy is as fast
_pinned_ring = {"ptr": None, "slots": 0, "chunk": 0, "lock": threading.Lock()}
_copy_pool = None


def _ring(lib):
    r = _pinned_ring
    if r["ptr"] is None:
        buf = C.c_void_p()
        _lib.check(lib.svmc_host_alloc(C.byref(buf), PIPELINE_SLOTS * PIPELINE_CHUNK_BYTES))
        r["ptr"], r["slots"], r["chunk"] = buf.value, PIPELINE_SLOTS, PIPELINE_CHUNK_BYTES
    return r


def pipelined_download(src_ptr: int, n_doubles: int, stream=None, out: Optional[np.ndarray] = None) -> np.ndarray:
    """n_doubles from device memory into `out` (a C-contiguous float64 array; a new one when None) through the pinned ring"""
    global _copy_pool
    lib = _lib.load()
    n = int(n_doubles)
    if out is None:
        out = np.empty(n, dtype=np.float64)
    flat = out.reshape(-1)
    if flat.size != n or flat.dtype != np.float64 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float64 array of the result's size")
    if 8 * n < PIPELINE_MIN_BYTES:
        _lib.check(lib.svmc_memcpy_d2h(flat.ctypes.data, src_ptr, 8 * n, stream))
        _lib.check(lib.svmc_stream_synchronize(stream))
        return out
    from concurrent.futures import ThreadPoolExecutor
    with _pinned_ring["lock"]:                     # one bulk download at a time per process: the ring is shared
        ring = _ring(lib)
        if _copy_pool is None:
            _copy_pool = ThreadPoolExecutor(max_workers=PIPELINE_THREADS, thread_name_prefix="svmc-d2h")
        per = ring["chunk"] // 8
        events, pending = [], [None] * ring["slots"]
        for _ in range(ring["slots"]):
            e = C.c_void_p()
            _lib.check(lib.svmc_event_create(C.byref(e)))
            events.append(e)

        def drain(slot, lo, cnt):
            _lib.check(lib.svmc_event_synchronize(events[slot]))
            view = np.ctypeslib.as_array(C.cast(ring["ptr"] + slot * ring["chunk"], C.POINTER(C.c_double)), shape=(cnt,))
            np.copyto(flat[lo:lo + cnt], view)

        try:
            for i, lo in enumerate(range(0, n, per)):
                slot, cnt = i % ring["slots"], min(per, n - lo)
                if pending[slot] is not None:
                    pending[slot].result()         # the slot's previous chunk has left the ring
                _lib.check(lib.svmc_memcpy_d2h(ring["ptr"] + slot * ring["chunk"], src_ptr + 8 * lo, 8 * cnt, stream))
                _lib.check(lib.svmc_event_record(events[slot], stream))
                pending[slot] = _copy_pool.submit(drain, slot, lo, cnt)
            for f in pending:
                if f is not None:
                    f.result()
        finally:
            # an exception inside the loop (a failed copy, a failed drain) must not destroy the events while drain tasks still
            # wait on them: every submitted task is waited for first, its own error swallowed -- the first one is already rising
            for f in pending:
                if f is not None:
                    try:
                        f.result()
                    except Exception:               # noqa: BLE001
                        pass
            for e in events:
                lib.svmc_event_destroy(e)
    return out


class DeviceArray:
    """a float64 [rows][cols] result left RESIDENT in HBM (simulate_vol_paths(..., return_device=True)): nothing crosses
    PCIe until asked.  Consumers: torch / cupy take it zero-copy (`torch.as_tensor(a, device="cuda")` through
    __cuda_array_interface__, `torch.from_dlpack(a)` through __dlpack__); `row_moments` / `expanding_mean_of_squares` do on
    the device what the reference's callers do with the host array; `numpy()` fetches it through the pinned pipeline."""

    def __init__(self, buf: DeviceBuffer, shape: Tuple[int, int], stream=None, device: int = 0, owns: bool = True, scratch=None):
        self._buf, self.shape, self.stream, self.device = buf, (int(shape[0]), int(shape[1])), stream, int(device)
        self.dtype = np.dtype(np.float64)
        self._owns = bool(owns)                    # False: a view of an engine's cached bulk buffer (free() leaves the buffer alone)
        self._scratch = scratch                    # callable(n_doubles, slot) -> DeviceBuffer for temporaries (the engine's cache)

    @property
    def ptr(self) -> int:
        if self._buf is None or self._buf.ptr is None:
            raise _lib.SvmcError("this DeviceArray was freed")
        return self._buf.ptr

    @property
    def nbytes(self) -> int:
        return 8 * self.shape[0] * self.shape[1]

    def synchronize(self) -> None:
        _lib.check(_lib.load().svmc_stream_synchronize(self.stream))

    def numpy(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        res = pipelined_download(self.ptr, self.shape[0] * self.shape[1], self.stream, None if out is None else out)
        return res.reshape(self.shape)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    @property
    def __cuda_array_interface__(self):
        self.synchronize()
        return {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 3, "strides": None}

    def __dlpack_device__(self):
        return (10, self.device)                   # kDLROCM

    def __dlpack__(self, stream=None, **_unused):
        self.synchronize()
        return _dlpack_capsule(self)

    def row_moments(self, center: float = 0.0, n_moments: int = 4) -> Tuple[np.ndarray, np.ndarray]:
        """(mean, std) over the paths, per row, of (a - center)^k for k = 1 .. n_moments: arrays [rows][n_moments];
        std is the population standard deviation (np.std): what moments_vol_qvar.py:48 computes from the host array"""
        lib = _lib.load()
        rows, cols = self.shape
        k2 = 2 * int(n_moments)
        if not 1 <= int(n_moments) <= 4:
            raise ValueError("n_moments must be 1 .. 4")
        sums, ws = DeviceBuffer(rows * k2), DeviceBuffer(rows * 4 * k2)
        try:
            _lib.check(lib.svmc_row_power_sums(self.ptr, cols, rows, cols, float(center), int(n_moments), sums.ptr, ws.ptr,
                                               ws.nbytes, self.stream))
            host = np.empty(rows * k2)
            _lib.check(lib.svmc_memcpy_d2h(host.ctypes.data, sums.ptr, host.nbytes, self.stream))
            self.synchronize()
        finally:
            sums.free()
            ws.free()
        s = host.reshape(k2, rows) / float(cols)               # s[j] = E[d^(j+1)]
        mean = np.stack([s[k] for k in range(n_moments)], axis=1)
        var = np.stack([s[2 * k + 1] - np.square(s[k]) for k in range(n_moments)], axis=1)
        return mean, np.sqrt(np.maximum(var, 0.0))

    def expanding_mean_of_squares(self) -> "DeviceArray":
        """a new resident array q[t][p] = mean of a[u][p]^2 over u <= t (moments_vol_qvar.py:102: the realised variance so far)"""
        rows, cols = self.shape
        if self._scratch is not None:              # a view of an engine's cache: the result lives in the engine's second bulk slot
            out, owns = self._scratch(rows * cols, 1), False
        else:
            out, owns = DeviceBuffer(rows * cols), True
        _lib.check(_lib.load().svmc_expanding_mean_squares(self.ptr, cols, rows, cols, out.ptr, cols, self.stream))
        return DeviceArray(out, self.shape, self.stream, self.device, owns=owns)

    def free(self) -> None:
        if self._buf is not None:
            self.synchronize()
            if self._owns:
                self._buf.free()
            self._buf = None


def _dlpack_capsule(arr: "DeviceArray"):
    """a DLPack capsule ("dltensor") of a DeviceArray: kDLROCM, float64, 2-d, compact strides; the capsule keeps the array
    alive until the consumer's deleter runs"""
    class DLDevice(C.Structure):
        _fields_ = [("device_type", C.c_int), ("device_id", C.c_int)]

    class DLDataType(C.Structure):
        _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]

    class DLTensor(C.Structure):
        _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int), ("dtype", DLDataType),
                    ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]

    class DLManagedTensor(C.Structure):
        pass
    DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
    DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p), ("deleter", DELETER)]
    shape = (C.c_int64 * 2)(*arr.shape)
    m = DLManagedTensor()
    m.dl_tensor = DLTensor(arr.ptr, DLDevice(10, arr.device), 2, DLDataType(2, 64, 1), shape, None, 0)
    key = id(m)

    @DELETER
    def deleter(_p):
        # called by the consumer THROUGH the ctypes thunk `deleter` itself, with `m` as its argument: dropping the last reference
        # to either from in here would free memory libffi / ctypes still read after this function returns.  The array (the HBM)
        # goes now; the thunk, the tensor struct and its shape retire and are released two capsule creations later
        kept = _DLPACK_ALIVE.pop(key, None)
        if kept is not None:
            _DLPACK_RETIRED[-1].append(kept[:3])
    m.deleter = deleter
    _DLPACK_RETIRED.append([])                             # a new generation; the one before the previous is let go
    del _DLPACK_RETIRED[:-2]
    _DLPACK_ALIVE[key] = (m, shape, deleter, arr)          # owner of everything the consumer may touch
    C.pythonapi.PyCapsule_New.restype = C.py_object
    C.pythonapi.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, _CAPSULE_DESTRUCTOR]
    return C.pythonapi.PyCapsule_New(C.addressof(m), b"dltensor", _drop_unconsumed_capsule)


_DLPACK_ALIVE = {}
_DLPACK_RETIRED = [[]]         # generations of (tensor struct, shape, deleter thunk) whose deleter has run (see _dlpack_capsule)
_CAPSULE_DESTRUCTOR = C.CFUNCTYPE(None, C.c_void_p)


@_CAPSULE_DESTRUCTOR
def _drop_unconsumed_capsule(capsule):
    """a capsule nobody consumed (still named "dltensor" when it dies) releases what it kept alive; a consumed one was renamed
    "used_dltensor" and its consumer calls the tensor's deleter"""
    C.pythonapi.PyCapsule_IsValid.restype = C.c_int
    C.pythonapi.PyCapsule_IsValid.argtypes = [C.c_void_p, C.c_char_p]
    if C.pythonapi.PyCapsule_IsValid(capsule, b"dltensor"):
        C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
        C.pythonapi.PyCapsule_GetPointer.argtypes = [C.c_void_p, C.c_char_p]
        addr = C.pythonapi.PyCapsule_GetPointer(capsule, b"dltensor")
        for key, kept in list(_DLPACK_ALIVE.items()):
            if C.addressof(kept[0]) == addr:
                _DLPACK_ALIVE.pop(key, None)


_CHAIN_CACHE: dict = {}


def marshalled_chain(ttms, forwards, discfactors, strikes: Sequence[np.ndarray], codes: Sequence[np.ndarray]) -> dict:
    """a chain's arrays as the fused C drivers take them (private contiguous copies, their ctypes pointers, the strike offsets and
    the slices that cut a result row into expiries), kept per chain CONTENT: a pricer is called on the same chain over and over
    (a calibration, a scenario sweep), and building these arrays costs as much as the launches of a small chain"""
    arrs = [np.asarray(ttms), np.asarray(forwards), np.asarray(discfactors)] + list(strikes) + list(codes)
    key = tuple((a.dtype.str, a.shape, a.tobytes()) for a in arrs)
    hit = _CHAIN_CACHE.get(key)
    if hit is not None:
        return hit
    m = len(strikes)
    dp = C.POINTER(C.c_double)
    f64 = lambda a: np.array(a, dtype=np.float64, order="C", copy=True).ravel()      # noqa: E731  (private copies)
    offs = np.concatenate([[0], np.cumsum([int(np.size(k)) for k in strikes])]).astype(np.uintp)
    total = int(offs[-1])
    t, f, d = f64(ttms), f64(forwards), f64(discfactors)
    if not (t.size == f.size == d.size == m == len(codes)):
        raise ValueError("chain arrays must have one entry per maturity")
    k_all = f64(np.concatenate([np.ravel(k) for k in strikes])) if total else np.zeros(1)
    c_all = np.array(np.concatenate([np.ravel(c) for c in codes]), dtype=np.int8) if total else np.zeros(1, dtype=np.int8)
    hit = {
        "keep": (t, f, d, k_all, c_all, offs), "total": total, "offs": offs, "m": m,
        "slices": [slice(int(offs[i]), int(offs[i + 1])) for i in range(m)],
        "ttms": t.ctypes.data_as(dp), "forwards": f.ctypes.data_as(dp), "discfactors": d.ctypes.data_as(dp),
        "strikes": k_all.ctypes.data_as(dp), "codes": c_all.ctypes.data_as(C.POINTER(C.c_int8)),
        "offsets": offs.ctypes.data_as(C.POINTER(C.c_size_t)),
    }
    if len(_CHAIN_CACHE) >= 16:
        _CHAIN_CACHE.clear()
    _CHAIN_CACHE[key] = hit
    return hit


BULK_KEEP_FRACTION = 0.125     # of the device's memory: what an engine's cached bulk buffers may hold between calls (trim_bulk)


class HipEngine:
    def __init__(self, n_path: int, device: Optional[int] = None, path_offset: int = 0,
                 n_snapshots: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        cnt = C.c_int()
        _lib.check(self.lib.svmc_device_count(C.byref(cnt)))
        if cnt.value < 1:
            raise _lib.SvmcError("no HIP device visible: the svmc Monte Carlo path runs on the GPU only")
        if device is not None:
            _lib.check(self.lib.svmc_set_device(int(device)))
        dev = C.c_int()
        _lib.check(self.lib.svmc_get_device(C.byref(dev)))
        self.device = dev.value
        self.n_path = int(n_path)
        self.path_offset = int(path_offset)
        self.stream = stream  # None = default stream
        self.x = DeviceBuffer(self.n_path)
        self.vol = DeviceBuffer(self.n_path)
        self.qvar = DeviceBuffer(self.n_path)
        ws = C.c_size_t()
        _lib.check(self.lib.svmc_slice_workspace_bytes(self.n_path, C.byref(ws)))
        self.ws_bytes = ws.value
        self.ws = DeviceBuffer(self.ws_bytes // 8)
        self._snap: Optional[DeviceBuffer] = None
        self._snap_rows = 0
        self._rand: Optional[DeviceBuffer] = None
        self._factors: Optional[DeviceBuffer] = None   # rough LogSV: [n_factors][n_path]
        self._sums = {}
        self._pinned, self._pinned_doubles = None, 0    # page-locked staging of the small result downloads
        self._bulk = {}                                 # slot -> DeviceBuffer: multi-GB results kept between calls (_bulk_buffer)
        self._device_bytes = None                       # the device's memory, asked once (trim_bulk)
        self._prof = None   # list of (name, start_event, stop_event) while kernel timing is on
        self._fused = None                              # svmc_session_t over this engine's state (fused_chain_session)
        self._fused_size = (0, 0)
        self._fused_timing = False
        self.closed = False
        if n_snapshots:
            self.reserve_snapshots(n_snapshots)

    # ---- plumbing -------------------------------------------------------------------------------
    def synchronize(self) -> None:
        _lib.check(self.lib.svmc_stream_synchronize(self.stream))

    def reserve_snapshots(self, rows: int) -> None:
        if self._snap is None or self._snap_rows < rows:
            if self._snap is not None:
                self._snap.free()
            self._snap = DeviceBuffer(rows * self.n_path)
            self._snap_rows = rows

    def snapshot_ptr(self, row: int) -> int:
        return self._snap.offset(row * self.n_path)

    def alloc_sums(self, n_doubles: int, tag: str = "sums") -> Tuple[int, object]:
        """named device buffer for reduction results; returns (ptr, owner).  Buffers with different tags
        never alias; a tag's buffer is re-used (grown) across calls."""
        buf = self._sums.get(tag)
        if buf is None or buf.n < n_doubles:
            if buf is not None:
                buf.free()
            buf = DeviceBuffer(max(int(n_doubles), 1))
            self._sums[tag] = buf
        return buf.ptr, buf

    PINNED_DOWNLOAD_MAX = 1 << 16       # doubles: the reduction results of a chain; bulk state goes the plain way

    def download(self, ptr: int, n: int) -> np.ndarray:
        """n doubles from device memory.  Small downloads (the payoff sums every chain call ends with) land in a page-locked
        buffer the engine keeps: a copy into pageable memory goes through the runtime's staging path and costs 16 us more
        per call than one into pinned memory (tools/ubench/sync_latency.py: 29.2 vs 16.3 us for kernel + copy + wait)."""
        n = int(n)
        if 0 < n <= self.PINNED_DOWNLOAD_MAX:
            if self._pinned is None or self._pinned_doubles < n:
                if self._pinned is not None:
                    _lib.check(self.lib.svmc_host_free(self._pinned))
                    self._pinned = None
                want = max(n, 4096)
                buf = C.c_void_p()
                _lib.check(self.lib.svmc_host_alloc(C.byref(buf), 8 * want))
                self._pinned, self._pinned_doubles = buf, want
            _lib.check(self.lib.svmc_memcpy_d2h(self._pinned, ptr, 8 * n, self.stream))
            self.synchronize()
            return np.ctypeslib.as_array(C.cast(self._pinned, C.POINTER(C.c_double)), shape=(n,)).copy()
        return pipelined_download(ptr, n, self.stream)      # bulk (terminal state vectors, path arrays): the pinned pipeline

    def upload(self, ptr: int, host: np.ndarray) -> None:
        host = np.ascontiguousarray(host, dtype=np.float64)
        _lib.check(self.lib.svmc_memcpy_h2d(ptr, host.ctypes.data, host.nbytes, self.stream))
        self.synchronize()  # the pageable source may be released by the caller

    # ---- kernel timing (HIP events on the launch stream; bench.py) ---------------------------------
    def start_kernel_timing(self) -> None:
        self._prof = []

    def _timed(self, name: str, launch) -> None:
        if self._prof is None:
            launch()
            return
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.svmc_event_create(C.byref(e0)))
        _lib.check(self.lib.svmc_event_create(C.byref(e1)))
        _lib.check(self.lib.svmc_event_record(e0, self.stream))
        launch()
        _lib.check(self.lib.svmc_event_record(e1, self.stream))
        self._prof.append((name, e0, e1))

    def stop_kernel_timing(self):
        """-> {kernel name: [durations in ms]} of every generator launch since start_kernel_timing()."""
        out = {}
        for name, e0, e1 in self._prof or []:
            if e1 is None:                               # a fused chain call: the session measured it (e0 holds the milliseconds)
                out.setdefault(name, []).append(e0)
                continue
            ms = C.c_float()
            _lib.check(self.lib.svmc_event_elapsed_ms(e0, e1, C.byref(ms)))
            out.setdefault(name, []).append(ms.value)
            self.lib.svmc_event_destroy(e0)
            self.lib.svmc_event_destroy(e1)
        self._prof = None
        return out

    # ---- one C-ABI call per chain (single GPU) -----------------------------------------------------
    def fused_chain_session(self, n_expiries: int, n_strikes: int):
        """a C-ABI session (svmc_session_create_on) over THIS engine's state arrays and stream, grown on demand: the fused
        chain drivers then leave the terminal state where get_state() reads it"""
        if self._fused is None or self._fused_size[0] < n_expiries or self._fused_size[1] < n_strikes:
            if self._fused is not None:
                _lib.check(self.lib.svmc_session_destroy(self._fused))
                self._fused = None
            size = (max(int(n_expiries), self._fused_size[0], 1), max(int(n_strikes), self._fused_size[1], 1))
            sess = C.c_void_p()
            _lib.check(self.lib.svmc_session_create_on(C.byref(sess), self.n_path, size[0], size[1], self.x.ptr, self.vol.ptr,
                                                       self.qvar.ptr, self.path_offset, self.stream))
            self._fused, self._fused_size, self._fused_timing = sess, size, False
        return self._fused

    def _fused_call(self, ch: dict, call, kernel_name: str):
        """run call(session, prices_ptr, stderrs_ptr) and cut the two result rows into per-expiry arrays; while kernel timing is
        on (start_kernel_timing) the session brackets its stepping launch with HIP events and the time is kept under kernel_name"""
        sess = self.fused_chain_session(ch["m"], ch["total"])
        timing = self._prof is not None
        if timing != self._fused_timing:
            _lib.check(self.lib.svmc_session_time_stepping(sess, int(timing)))
            self._fused_timing = timing
        res = np.empty((2, max(ch["total"], 1)))
        dp = C.POINTER(C.c_double)
        _lib.check(call(sess, C.cast(res.ctypes.data, dp), C.cast(res.ctypes.data + res.strides[0], dp)))
        if timing:
            ms = C.c_float()
            _lib.check(self.lib.svmc_session_last_stepping_ms(sess, C.byref(ms)))
            self._prof.append((kernel_name, float(ms.value), None))
        return [res[0, sl] for sl in ch["slices"]], [res[1, sl] for sl in ch["slices"]]

    def price_logsv_chain_fused(self, ch: dict, v0, theta, kappa1, kappa2, beta, volvol, etas, is_spot_measure, nb_steps_per_year,
                                variable_type: int, seed: int, call_id: int):
        """logsv_mc_chain_pricer on one GPU as ONE svmc_logsv_chain_price call on this engine's state (ch: marshalled_chain):
        the kernels, their order and their arguments are those of mc_chain.price_chain_on_engine -- the same bits -- without
        the dozen ctypes calls and NumPy temporaries around them (tools/r06/chain_call_breakdown.py: 0.135 -> 0.114 ms for a
        4 x 13 chain at 2^16 paths)"""
        etas = np.ascontiguousarray(etas, dtype=np.float64)
        if etas.size != ch["m"]:
            raise ValueError("vol_backbone_etas must have one entry per maturity")
        return self._fused_call(ch, lambda sess, p, e: self.lib.svmc_logsv_chain_price(
            sess, ch["ttms"], ch["forwards"], ch["discfactors"], etas.ctypes.data_as(C.POINTER(C.c_double)), ch["m"], ch["strikes"],
            ch["codes"], ch["offsets"], float(v0), float(theta), float(kappa1), float(kappa2), float(beta), float(volvol),
            int(bool(is_spot_measure)), int(nb_steps_per_year), int(variable_type), int(seed), int(call_id), p, e),
            "logsv_rng_kernel" if ch["m"] == 1 else "logsv_chain_rng_kernel")

    def price_heston_chain_fused(self, ch: dict, v0, theta, kappa, rho, volvol, scheme: int, nb_steps_per_year, variable_type: int,
                                 seed: int, call_id: int):
        """heston_mc_chain_pricer on one GPU as ONE svmc_heston_chain_price call on this engine's state"""
        return self._fused_call(ch, lambda sess, p, e: self.lib.svmc_heston_chain_price(
            sess, ch["ttms"], ch["forwards"], ch["discfactors"], ch["m"], ch["strikes"], ch["codes"], ch["offsets"], float(v0),
            float(theta), float(kappa), float(rho), float(volvol), int(scheme), int(nb_steps_per_year), int(variable_type),
            int(seed), int(call_id), p, e), "heston_rng_kernel" if ch["m"] == 1 else "heston_chain_rng_kernel")

    # ---- state ----------------------------------------------------------------------------------
    def fill_state(self, x0: float, vol0: float, qvar0: float) -> None:
        _lib.check(self.lib.svmc_fill_state(self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path,
                                            float(x0), float(vol0), float(qvar0), self.stream))

    def set_state(self, x: np.ndarray, vol: np.ndarray, qvar: np.ndarray) -> None:
        for buf, a in ((self.x, x), (self.vol, vol), (self.qvar, qvar)):
            a = np.ascontiguousarray(a, dtype=np.float64)
            assert a.shape == (self.n_path,)
            self.upload(buf.ptr, a)

    def get_state(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        return (self.download(self.x.ptr, self.n_path), self.download(self.vol.ptr, self.n_path),
                self.download(self.qvar.ptr, self.n_path))

    def snapshot(self, row: int, which: str = "x") -> None:
        src = self.x.ptr if which == "x" else self.qvar.ptr
        _lib.check(self.lib.svmc_memcpy_d2d(self.snapshot_ptr(row), src, 8 * self.n_path, self.stream))

    def _bulk_buffer(self, n_doubles: int, slot: int = 0) -> DeviceBuffer:
        """the engine's cached buffer for a bulk result that does not leave the call (the path array behind a NumPy return, the
        arrays vol_path_moments reduces): grown when needed, kept until close() / release_bulk().  Allocating and freeing 8.6 GB
        per call costs the runtime 0.25-0.7 s every other call (the VRAM is mapped and unmapped; profiles/r05_moments_timing
        .jsonl) -- two hundred times the kernel that fills it."""
        buf = self._bulk.get(slot)
        if buf is None or buf.n < n_doubles:
            if buf is not None:
                buf.free()
            buf = DeviceBuffer(int(n_doubles))
            self._bulk[slot] = buf
        return buf

    def trim_bulk(self) -> None:
        """after a call whose bulk result has left the device: keep the cached buffers only while together they stay under an
        eighth of the device's memory (36 GB of an MI355X's 288: C2-sized path arrays, 8.6 GB each, stay; a caller that once
        simulated 10^7 paths does not pin 80 GB for the life of the process) -- BULK_KEEP_FRACTION, release_bulk() for all of it"""
        held = 8 * sum(b.n for b in self._bulk.values())
        if held == 0:
            return
        if self._device_bytes is None:
            name, cus, khz, mem = C.create_string_buffer(128), C.c_int(), C.c_int(), C.c_size_t()
            _lib.check(self.lib.svmc_device_info(self.device, name, 128, C.byref(cus), C.byref(khz), C.byref(mem)))
            self._device_bytes = int(mem.value)
        if held > BULK_KEEP_FRACTION * self._device_bytes:
            self.release_bulk()

    def release_bulk(self) -> None:
        """give the cached bulk buffers back (they otherwise stay with the engine: up to two path arrays of the largest size seen)"""
        self.synchronize()
        for b in self._bulk.values():
            b.free()
        self._bulk = {}

    # ---- randoms --------------------------------------------------------------------------------
    def _rand_buffer(self, n_doubles: int) -> DeviceBuffer:
        if self._rand is None or self._rand.n < n_doubles:
            if self._rand is not None:
                self._rand.free()
            self._rand = DeviceBuffer(n_doubles)
        return self._rand

    def upload_randoms(self, arrays: Sequence[np.ndarray], col0: int = 0) -> Tuple[int, ...]:
        """upload the local column range [col0, col0 + n_path) of host [nb_steps, nb_path_total] arrays."""
        nb = arrays[0].shape[0]
        buf = self._rand_buffer(len(arrays) * nb * self.n_path)
        ptrs, keep = [], []         # `keep`: converted temporaries must outlive the asynchronous copies below
        for i, a in enumerate(arrays):
            a = np.asarray(a)
            if a.dtype != np.float64 or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            assert a.ndim == 2 and a.shape[0] == nb and a.shape[1] >= col0 + self.n_path
            dst = buf.offset(i * nb * self.n_path)
            _lib.check(self.lib.svmc_memcpy2d_h2d(dst, 8 * self.n_path, a.ctypes.data + 8 * col0, 8 * a.shape[1],
                                                  8 * self.n_path, nb, self.stream))
            ptrs.append(dst)
        self.synchronize()
        del keep
        return tuple(ptrs)

    def fill_normals(self, nb_steps: int, seed: int, call_id: int = 0, step_offset: int = 0) -> Tuple[int, int]:
        buf = self._rand_buffer(2 * nb_steps * self.n_path)
        w0, w1 = buf.ptr, buf.offset(nb_steps * self.n_path)
        _lib.check(self.lib.svmc_fill_normals(w0, w1, self.n_path, self.n_path, nb_steps, seed, call_id,
                                              self.path_offset, step_offset, self.stream))
        return w0, w1

    # ---- generators -----------------------------------------------------------------------------
    def logsv_rng(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, seed, call_id,
                  step_offset) -> None:
        self._timed("logsv_rng_kernel", lambda: _lib.check(self.lib.svmc_logsv_terminal_rng(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa1), float(kappa2), float(beta), float(volvol), float(eta), int(bool(is_spot_measure)),
            int(seed), int(call_id), self.path_offset, int(step_offset), self.stream)))

    def logsv_slice_rng(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, seed, call_id,
                        step_offset, forward, snap_row, qvar_row, spot_ptr, start=None) -> None:
        """advance + snapshot + spot sums of one expiry in one kernel (svmc_logsv_slice_rng).  start = (x0, sigma0, qvar0):
        every path starts there (svmc_logsv_slice_rng_from: no fill launch, the state buffers are outputs only)"""
        fn = self.lib.svmc_logsv_slice_rng if start is None else functools.partial(self.lib.svmc_logsv_slice_rng_from,
                                                                                   *[float(v) for v in start])
        self._timed("logsv_rng_kernel", lambda: _lib.check(fn(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa1), float(kappa2), float(beta), float(volvol), float(eta), int(bool(is_spot_measure)),
            int(seed), int(call_id), self.path_offset, int(step_offset), float(forward), self.snapshot_ptr(snap_row),
            None if qvar_row is None else self.snapshot_ptr(qvar_row), spot_ptr, self.ws.ptr, self.ws_bytes,
            self.stream)))

    def logsv_chain_rng(self, nb_steps: Sequence[int], dts: Sequence[float], etas: Sequence[float],
                        forwards: Sequence[float], theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id,
                        step_offset, need_qvar: bool, spot_ptr: int, start=None) -> None:
        """every expiry of a chain in one stepping launch (svmc_logsv_chain_rng): snapshot rows 0..m-1 get the
        terminal x of each expiry, rows m..2m-1 the quadratic variance (need_qvar), spot_ptr the 2m spot sums;
        start = (x0, sigma0, qvar0) as in logsv_slice_rng"""
        m = len(nb_steps)
        dp = C.POINTER(C.c_double)
        nbs = (C.c_int * m)(*[int(v) for v in nb_steps])
        d, e, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (dts, etas, forwards))
        fn = self.lib.svmc_logsv_chain_rng if start is None else functools.partial(self.lib.svmc_logsv_chain_rng_from,
                                                                                   *[float(v) for v in start])
        self._timed("logsv_chain_rng_kernel", lambda: _lib.check(fn(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, m, nbs, d.ctypes.data_as(dp), e.ctypes.data_as(dp),
            f.ctypes.data_as(dp), float(theta), float(kappa1), float(kappa2), float(beta), float(volvol),
            int(bool(is_spot_measure)), int(seed), int(call_id), self.path_offset, int(step_offset), self.snapshot_ptr(0),
            self.snapshot_ptr(m) if need_qvar else None, spot_ptr, self.ws.ptr, self.ws_bytes, self.stream)))

    def heston_chain_rng(self, nb_steps: Sequence[int], dts: Sequence[float], forwards: Sequence[float], theta, kappa,
                         rho, volvol, scheme, seed, call_id, step_offset, need_qvar: bool, spot_ptr: int, start=None) -> None:
        """every expiry of a Heston chain in one stepping launch (svmc_heston_chain_rng); layout and `start` as
        logsv_chain_rng"""
        m = len(nb_steps)
        dp = C.POINTER(C.c_double)
        nbs = (C.c_int * m)(*[int(v) for v in nb_steps])
        d, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (dts, forwards))
        fn = self.lib.svmc_heston_chain_rng if start is None else functools.partial(self.lib.svmc_heston_chain_rng_from,
                                                                                    *[float(v) for v in start])
        self._timed("heston_chain_rng_kernel", lambda: _lib.check(fn(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, m, nbs, d.ctypes.data_as(dp), f.ctypes.data_as(dp),
            float(theta), float(kappa), float(rho), float(volvol), int(scheme), int(seed), int(call_id), self.path_offset,
            int(step_offset), self.snapshot_ptr(0), self.snapshot_ptr(m) if need_qvar else None, spot_ptr, self.ws.ptr,
            self.ws_bytes, self.stream)))

    def heston_slice_rng(self, nb_steps, dt, theta, kappa, rho, volvol, scheme, seed, call_id, step_offset, forward,
                         snap_row, qvar_row, spot_ptr, start=None) -> None:
        fn = self.lib.svmc_heston_slice_rng if start is None else functools.partial(self.lib.svmc_heston_slice_rng_from,
                                                                                    *[float(v) for v in start])
        self._timed("heston_rng_kernel", lambda: _lib.check(fn(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa), float(rho), float(volvol), int(scheme), int(seed), int(call_id), self.path_offset,
            int(step_offset), float(forward), self.snapshot_ptr(snap_row),
            None if qvar_row is None else self.snapshot_ptr(qvar_row), spot_ptr, self.ws.ptr, self.ws_bytes,
            self.stream)))

    def finish_slice(self, forward, snap_row, qvar_row, spot_ptr) -> None:
        """the un-fused equivalent for generators without a slice epilogue (streamed randoms)"""
        self.snapshot(snap_row, "x")
        if qvar_row is not None:
            self.snapshot(qvar_row, "qvar")
        self.spot_sums(self.snapshot_ptr(snap_row), forward, spot_ptr)

    def logsv_w(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, w0_ptr, w1_ptr,
                ldw=None) -> None:
        self._timed("logsv_w_kernel", lambda: _lib.check(self.lib.svmc_logsv_terminal_w(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa1), float(kappa2), float(beta), float(volvol), float(eta), int(bool(is_spot_measure)),
            w0_ptr, w1_ptr, self.n_path if ldw is None else int(ldw), self.stream)))

    def rough_logsv(self, nb_steps, h, nodes, weights, v0, theta, kappa1, kappa2, rho, volvol, z0_ptr=None,
                    z1_ptr=None, ldw=None, seed=0, call_id=0, step_offset=0, from_origin=True, slice_out=None) -> None:
        """rough LogSV terminal state in (x, factor buffer, qvar); randoms streamed from z0/z1 or drawn on device.
        slice_out = (forward, snap_row, qvar_row | None, spot_ptr) adds the slice epilogue (svmc_rough_logsv_slice)."""
        nodes, weights, v0 = (np.ascontiguousarray(a, dtype=np.float64) for a in (nodes, weights, v0))
        if not (nodes.ndim == 1 and nodes.shape == weights.shape == v0.shape):
            raise ValueError("nodes, weights and v0 must be 1-d arrays of one length")
        if self._factors is None:
            self._factors = DeviceBuffer(3 * self.n_path)
        dp = C.POINTER(C.c_double)
        args = (self.x.ptr, self._factors.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(h), int(nodes.size),
                nodes.ctypes.data_as(dp), weights.ctypes.data_as(dp), v0.ctypes.data_as(dp), float(theta), float(kappa1),
                float(kappa2), float(rho), float(volvol), z0_ptr, z1_ptr, self.n_path if ldw is None else int(ldw),
                int(seed), int(call_id), self.path_offset, int(step_offset), int(bool(from_origin)))
        if slice_out is None:
            self._timed("rough_logsv_kernel",
                        lambda: _lib.check(self.lib.svmc_rough_logsv_terminal(*args, self.stream)))
            return
        forward, snap_row, qvar_row, spot_ptr = slice_out
        self._timed("rough_logsv_kernel", lambda: _lib.check(self.lib.svmc_rough_logsv_slice(
            *args, float(forward), self.snapshot_ptr(snap_row), None if qvar_row is None else self.snapshot_ptr(qvar_row),
            spot_ptr, self.ws.ptr, self.ws_bytes, self.stream)))

    def rough_logsv_chain(self, nb_steps: Sequence[int], hs: Sequence[float], forwards: Sequence[float], nodes, weights, v0,
                          theta, kappa1, kappa2, rho, volvol, need_qvar: bool, spot_ptr: int, z0_ptr=None, z1_ptr=None,
                          ldw=None, seed=0, call_id=0) -> None:
        """every expiry of a rough-LogSV chain in one stepping launch (svmc_rough_logsv_chain): each expiry simulated from
        time 0 on its own step, side by side; snapshot rows 0..m-1 get the terminal x, rows m..2m-1 the quadratic
        variance (need_qvar), spot_ptr the 2m spot sums"""
        nodes, weights, v0 = (np.ascontiguousarray(a, dtype=np.float64) for a in (nodes, weights, v0))
        if not (nodes.ndim == 1 and nodes.shape == weights.shape == v0.shape):
            raise ValueError("nodes, weights and v0 must be 1-d arrays of one length")
        if self._factors is None:
            self._factors = DeviceBuffer(3 * self.n_path)
        m = len(nb_steps)
        dp = C.POINTER(C.c_double)
        nbs = (C.c_int * m)(*[int(v) for v in nb_steps])
        h, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (hs, forwards))
        self._timed("rough_logsv_expiries_kernel", lambda: _lib.check(self.lib.svmc_rough_logsv_chain(
            self.x.ptr, self._factors.ptr, self.qvar.ptr, self.n_path, m, nbs, h.ctypes.data_as(dp), f.ctypes.data_as(dp),
            int(nodes.size), nodes.ctypes.data_as(dp), weights.ctypes.data_as(dp), v0.ctypes.data_as(dp), float(theta),
            float(kappa1), float(kappa2), float(rho), float(volvol), z0_ptr, z1_ptr, self.n_path if ldw is None else int(ldw),
            int(seed), int(call_id), self.path_offset, self.snapshot_ptr(0), self.snapshot_ptr(m) if need_qvar else None,
            spot_ptr, self.ws.ptr, self.ws_bytes, self.stream)))

    def get_factors(self, n_factors: int) -> np.ndarray:
        return self.download(self._factors.ptr, n_factors * self.n_path).reshape(n_factors, self.n_path)

    def logsv_slice_w(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, w0_ptr, w1_ptr,
                      forward, snap_row, qvar_row, spot_ptr, ldw=None) -> None:
        """streamed-randoms advance + snapshot + spot sums of one expiry in one stepping launch (svmc_logsv_slice_w)"""
        self._timed("logsv_w_kernel", lambda: _lib.check(self.lib.svmc_logsv_slice_w(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa1), float(kappa2), float(beta), float(volvol), float(eta), int(bool(is_spot_measure)),
            w0_ptr, w1_ptr, self.n_path if ldw is None else int(ldw), float(forward), self.snapshot_ptr(snap_row),
            None if qvar_row is None else self.snapshot_ptr(qvar_row), spot_ptr, self.ws.ptr, self.ws_bytes,
            self.stream)))

    def logsv_vol_paths(self, nb_steps, dt, v0, theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id,
                        brownians: Optional[np.ndarray] = None, return_device=False, out_host: Optional[np.ndarray] = None):
        """full-grid sigma paths [(nb_steps+1), n_path] (svmc_logsv_vol_paths): the host array (fetched through the pinned
        pipeline, into out_host when given); with return_device=True the resident DeviceArray (a fresh allocation the caller
        owns and frees); with return_device="view" a DeviceArray over the engine's cached bulk buffer (valid until the engine's
        next bulk call; what vol_path_moments reduces)"""
        n_out = (nb_steps + 1) * self.n_path
        out = DeviceBuffer(n_out) if return_device is True else self._bulk_buffer(n_out, 0)
        b_ptr = None
        if brownians is not None:
            (b_ptr,) = self.upload_randoms((brownians,))
        _lib.check(self.lib.svmc_logsv_vol_paths(out.ptr, self.n_path, self.n_path, int(nb_steps), float(dt), float(v0),
                                                 float(theta), float(kappa1), float(kappa2), float(beta), float(volvol),
                                                 int(bool(is_spot_measure)), b_ptr, self.n_path, int(seed),
                                                 int(call_id), self.path_offset, self.stream))
        arr = DeviceArray(out, (nb_steps + 1, self.n_path), self.stream, self.device, owns=return_device is True,
                          scratch=None if return_device is True else self._bulk_buffer)
        if return_device:
            return arr
        host = arr.numpy(out_host)
        arr.free()
        self.trim_bulk()
        return host

    def heston_rng(self, nb_steps, dt, theta, kappa, rho, volvol, scheme, seed, call_id, step_offset) -> None:
        self._timed("heston_rng_kernel", lambda: _lib.check(self.lib.svmc_heston_terminal_rng(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa), float(rho), float(volvol), int(scheme), int(seed), int(call_id), self.path_offset,
            int(step_offset), self.stream)))

    def heston_w(self, nb_steps, dt, theta, kappa, rho, volvol, w0_ptr, w1_ptr, ldw=None) -> None:
        _lib.check(self.lib.svmc_heston_terminal_w(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa), float(rho), float(volvol), w0_ptr, w1_ptr,
            self.n_path if ldw is None else int(ldw), self.stream))

    def heston_qe_w(self, nb_steps, dt, theta, kappa, rho, volvol, z0_ptr, z1_ptr, u_ptr, ldw=None) -> None:
        _lib.check(self.lib.svmc_heston_qe_terminal_w(
            self.x.ptr, self.vol.ptr, self.qvar.ptr, self.n_path, int(nb_steps), float(dt), float(theta),
            float(kappa), float(rho), float(volvol), z0_ptr, z1_ptr, u_ptr,
            self.n_path if ldw is None else int(ldw), self.stream))

    # ---- payoff reduction -----------------------------------------------------------------------
    def spot_sums(self, x_ptr: int, forward: float, out_ptr: int) -> None:
        _lib.check(self.lib.svmc_spot_sums(x_ptr, self.n_path, float(forward), out_ptr, self.ws.ptr,
                                           self.ws_bytes, self.stream))

    def payoff_sums(self, x_ptr: int, qvar_ptr: Optional[int], forward: float, ttm: float, spot_sums_ptr: int,
                    strikes: np.ndarray, codes: np.ndarray, shifts: np.ndarray, variable_type: int,
                    out_ptr: int) -> None:
        k = len(strikes)
        _lib.check(self.lib.svmc_payoff_sums(
            x_ptr, qvar_ptr, self.n_path, float(forward), float(ttm), spot_sums_ptr,
            strikes.ctypes.data_as(C.POINTER(C.c_double)), codes.ctypes.data_as(C.POINTER(C.c_int8)),
            shifts.ctypes.data_as(C.POINTER(C.c_double)), k, int(variable_type), out_ptr, self.ws.ptr,
            self.ws_bytes, self.stream))

    def payoff_sums_chain(self, snap_rows: Sequence[int], qvar_rows: Optional[Sequence[int]], forwards, ttms,
                          spot_sums_ptr: int, strikes: Sequence[np.ndarray], codes: Sequence[np.ndarray],
                          shifts: Sequence[np.ndarray], variable_type: int, out_ptr: int) -> None:
        """per-strike payoff sums of ALL expiries in one pair of launches (svmc_payoff_sums_chain); expiry i reads
        snapshot row snap_rows[i] (and qvar_rows[i]) and the recentring sums at spot_sums_ptr + 16 i"""
        m = len(strikes)
        offs = np.concatenate([[0], np.cumsum([len(k) for k in strikes])]).astype(np.uintp)
        if int(offs[-1]) == 0:
            return
        dp = C.POINTER(C.c_double)
        k_all = np.ascontiguousarray(np.concatenate(strikes), dtype=np.float64)
        c_all = np.ascontiguousarray(np.concatenate(codes), dtype=np.int8)
        s_all = np.ascontiguousarray(np.concatenate(shifts), dtype=np.float64)
        fw = np.ascontiguousarray(forwards, dtype=np.float64)
        tt = np.ascontiguousarray(ttms, dtype=np.float64)
        xs = (C.c_void_p * m)(*[self.snapshot_ptr(r) for r in snap_rows])
        qs = None if qvar_rows is None else (C.c_void_p * m)(*[self.snapshot_ptr(r) for r in qvar_rows])
        _lib.check(self.lib.svmc_payoff_sums_chain(
            xs, qs, self.n_path, fw.ctypes.data_as(dp), tt.ctypes.data_as(dp), spot_sums_ptr, m, k_all.ctypes.data_as(dp),
            c_all.ctypes.data_as(C.POINTER(C.c_int8)), s_all.ctypes.data_as(dp), offs.ctypes.data_as(C.POINTER(C.c_size_t)),
            int(variable_type), out_ptr, self.ws.ptr, self.ws_bytes, self.stream))

    def close(self) -> None:
        """free every HBM buffer of the engine; any later launch through it fails in the C ABI's null-pointer
        checks (SvmcError), it never touches freed memory"""
        self.closed = True
        if self._fused is not None:                 # the session borrows x / vol / qvar: it goes first
            self.lib.svmc_session_destroy(self._fused)
            self._fused, self._fused_size = None, (0, 0)
        self.__dict__.pop("_comm_bufs", None)       # a communicator's reduction tensors that lived on this engine (dist.TorchComm)
        for b in (self.x, self.vol, self.qvar, self.ws, self._snap, self._rand, self._factors, *self._sums.values(),
                  *self._bulk.values()):
            if b is not None:
                b.free()
        if self._pinned is not None:
            self.lib.svmc_host_free(self._pinned)
            self._pinned, self._pinned_doubles = None, 0


class DeviceRandoms:
    """fixed randoms of a chain kept resident in HBM: upload once, re-price at kernel speed on
    every calibration iterate.  Holds, per expiry, this rank's column range of W0 and W1 ([nb_steps_i][n_local])."""

    def __init__(self, W0s: Sequence[np.ndarray], W1s: Sequence[np.ndarray], dts: Sequence[float], n_local: int,
                 col0: int = 0):
        lib = _lib.load()
        self.nb_path = int(np.asarray(W0s[0]).shape[1])
        self.n_local, self.col0 = int(n_local), int(col0)
        self.dts = [float(d) for d in dts]
        self.nb_steps, self.w0, self.w1 = [], [], []
        self._session, self._session_strikes = None, 0
        self._session_sets, self._session_sets_size = None, (0, 0)
        for W0, W1 in zip(W0s, W1s):
            W0 = np.ascontiguousarray(W0, dtype=np.float64)
            W1 = np.ascontiguousarray(W1, dtype=np.float64)
            if W0.shape != W1.shape or W0.shape[1] != self.nb_path:
                raise ValueError("every W0/W1 must have shape [nb_steps_i, nb_path]")
            nb = W0.shape[0]
            bufs = []
            for a in (W0, W1):
                buf = DeviceBuffer(nb * self.n_local)
                _lib.check(lib.svmc_memcpy2d_h2d(buf.ptr, 8 * self.n_local, a.ctypes.data + 8 * self.col0,
                                                 8 * a.shape[1], 8 * self.n_local, nb, None))
                bufs.append(buf)
            _lib.check(lib.svmc_stream_synchronize(None))
            self.nb_steps.append(nb)
            self.w0.append(bufs[0])
            self.w1.append(bufs[1])

    @classmethod
    def drawn_on_device(cls, nb_steps: Sequence[int], dts: Sequence[float], nb_path: int, n_local: int, col0: int,
                        seed: int, call_id: int = 0) -> "DeviceRandoms":
        """the chain's fixed randoms drawn IN HBM by the counter-based generator (svmc_fill_normals) instead of being
        drawn by NumPy and uploaded: expiry i holds the normals the on-device-RNG generators consume for
        (seed, call_id) at steps sum(nb_steps[:i]) .. -- so a chain priced on them is the on-device-RNG chain with its
        randoms frozen, and a calibration no longer starts with a second of host Mersenne-Twister draws."""
        lib = _lib.load()
        self = cls.__new__(cls)
        self.nb_path, self.n_local, self.col0 = int(nb_path), int(n_local), int(col0)
        self.dts = [float(d) for d in dts]
        self.nb_steps, self.w0, self.w1 = [int(n) for n in nb_steps], [], []
        self._session, self._session_strikes = None, 0
        self._session_sets, self._session_sets_size = None, (0, 0)
        step0 = 0
        for nb in self.nb_steps:
            b0, b1 = DeviceBuffer(nb * self.n_local), DeviceBuffer(nb * self.n_local)
            _lib.check(lib.svmc_fill_normals(b0.ptr, b1.ptr, self.n_local, self.n_local, nb, int(seed), int(call_id),
                                             self.col0, step0, None))
            self.w0.append(b0)
            self.w1.append(b1)
            step0 += nb
        _lib.check(lib.svmc_stream_synchronize(None))
        return self

    @classmethod
    def frozen(cls, nb_steps: Sequence[int], dts: Sequence[float], nb_path: int, n_local: int, col0: int, seed: int,
               call_id: int = 0) -> "DeviceRandoms":
        """the chain's fixed randoms as NOTHING BUT their definition: the counter-based stream of (seed, call_id).  No array
        exists, on the host or in HBM -- every pricing regenerates the same normals in registers (svmc_logsv_chain_price_
        frozen_sets), so a chain priced on this object is logsv_mc_chain_pricer(seed=seed) with the given call id, bit for
        bit, at every call.  The reference's MC calibration keeps RandomState arrays for the same purpose
        (pricers/logsv_pricer.py:244-265, 520-527); drawn_on_device() kept 16 bytes per path-step of HBM."""
        self = cls.__new__(cls)
        self.nb_path, self.n_local, self.col0 = int(nb_path), int(n_local), int(col0)
        self.dts = [float(d) for d in dts]
        self.nb_steps, self.w0, self.w1 = [int(n) for n in nb_steps], [], []
        self._session, self._session_strikes = None, 0
        self._session_sets, self._session_sets_size = None, (0, 0)
        self.frozen_stream = (int(seed), int(call_id))
        return self

    @property
    def is_frozen(self) -> bool:
        return getattr(self, "frozen_stream", None) is not None

    def __len__(self):
        return len(self.nb_steps)

    def free(self) -> None:
        for b in self.w0 + self.w1:
            b.free()
        if self._session is not None:
            _lib.check(_lib.load().svmc_session_destroy(self._session))
            self._session = None
        if self._session_sets is not None:
            _lib.check(_lib.load().svmc_session_destroy(self._session_sets))
            self._session_sets = None

    def graph_launches(self) -> int:
        """how many chain pricings of this object were hipGraph replays (diagnostics)"""
        if self._session is None:
            return 0
        n = C.c_size_t()
        _lib.check(_lib.load().svmc_session_graph_launches(self._session, C.byref(n)))
        return int(n.value)

    def price_logsv_chain(self, ttms, forwards, discfactors, strikes: Sequence[np.ndarray], codes: Sequence[np.ndarray],
                          v0, theta, kappa1, kappa2, beta, volvol, etas, is_spot_measure: bool, variable_type: int,
                          use_graph: bool = True, want_ivols: bool = False):
        """one call of the fused single-GPU driver svmc_logsv_chain_price_fixed on these randoms: the chain's launches
        captured once into a hipGraph and replayed per parameter set (use_graph=False: queued back to back instead),
        one synchronisation, prices and stderrs back -- the inner loop of an MC calibration.  Same kernels in the
        same order as mc_chain.price_chain_on_engine, hence the same bits.  want_ivols: a third list, the Black-76
        implied vols of the prices, computed by the graph's last kernel (svmc_logsv_chain_price_fixed_iv)."""
        if self.is_frozen:
            row = np.concatenate([[v0, theta, kappa1, kappa2, beta, volvol], np.asarray(etas, dtype=np.float64).ravel()])
            return self.price_logsv_chain_sets(ttms, forwards, discfactors, strikes, codes, row[None, :], is_spot_measure,
                                               variable_type, want_ivols=want_ivols, use_graph=use_graph)[0]
        lib = _lib.load()
        m = len(self)
        offs = np.concatenate([[0], np.cumsum([len(k) for k in strikes])]).astype(np.uintp)
        total = int(offs[-1])
        if self._session is None or self._session_strikes < total:
            if self._session is not None:
                _lib.check(lib.svmc_session_destroy(self._session))
            sess = C.c_void_p()
            _lib.check(lib.svmc_session_create(C.byref(sess), self.n_local, m, max(total, 1)))
            self._session, self._session_strikes = sess, max(total, 1)
        _lib.check(lib.svmc_session_use_graphs(self._session, int(bool(use_graph))))
        dp = C.POINTER(C.c_double)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)      # noqa: E731
        ttms, forwards, discfactors, etas = f64(ttms), f64(forwards), f64(discfactors), f64(etas)
        k_all = f64(np.concatenate(strikes)) if total else np.zeros(1)
        c_all = np.ascontiguousarray(np.concatenate(codes), dtype=np.int8) if total else np.zeros(1, dtype=np.int8)
        w0 = (C.c_void_p * m)(*[b.ptr for b in self.w0])
        w1 = (C.c_void_p * m)(*[b.ptr for b in self.w1])
        nbs = (C.c_int * m)(*self.nb_steps)
        dts = f64(self.dts)
        prices, stderrs = np.empty(max(total, 1)), np.empty(max(total, 1))
        ivols = np.empty(max(total, 1)) if want_ivols else None
        _lib.check(lib.svmc_logsv_chain_price_fixed_iv(
            self._session, ttms.ctypes.data_as(dp), forwards.ctypes.data_as(dp), discfactors.ctypes.data_as(dp),
            etas.ctypes.data_as(dp), m, k_all.ctypes.data_as(dp), c_all.ctypes.data_as(C.POINTER(C.c_int8)),
            offs.ctypes.data_as(C.POINTER(C.c_size_t)), float(v0), float(theta), float(kappa1), float(kappa2),
            float(beta), float(volvol), int(bool(is_spot_measure)), int(variable_type), w0, w1, nbs,
            dts.ctypes.data_as(dp), self.n_local, prices.ctypes.data_as(dp), stderrs.ctypes.data_as(dp),
            ivols.ctypes.data_as(dp) if want_ivols else None))
        split = lambda a: [a[offs[i]:offs[i + 1]].copy() for i in range(m)]      # noqa: E731
        return (split(prices), split(stderrs), split(ivols)) if want_ivols else (split(prices), split(stderrs))


    def _marshalled_chain(self, ttms, forwards, discfactors, strikes, codes):
        """the chain's arrays as the C ABI takes them (contiguous copies, their ctypes pointers, the strike offsets and the
        slices that cut a result row into expiries), kept per chain CONTENT: the objective of a calibration prices the same
        chain hundreds of times and the marshalling was a quarter of an evaluation's wall time."""
        arrs = [np.asarray(ttms), np.asarray(forwards), np.asarray(discfactors)] + list(strikes) + list(codes)
        key = tuple((a.dtype.str, a.shape, a.tobytes()) for a in arrs)
        cache = self.__dict__.setdefault("_chain_cache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        m = len(self)
        dp = C.POINTER(C.c_double)
        f64 = lambda a: np.array(a, dtype=np.float64, order="C", copy=True).ravel()      # noqa: E731  (private copies)
        offs = np.concatenate([[0], np.cumsum([len(k) for k in strikes])]).astype(np.uintp)
        total = int(offs[-1])
        t, f, d = f64(ttms), f64(forwards), f64(discfactors)
        k_all = f64(np.concatenate(strikes)) if total else np.zeros(1)
        c_all = np.array(np.concatenate(codes), dtype=np.int8) if total else np.zeros(1, dtype=np.int8)
        dts = f64(self.dts)
        hit = {
            "keep": (t, f, d, k_all, c_all, offs, dts), "total": total, "offs": offs,
            "slices": [slice(int(offs[i]), int(offs[i + 1])) for i in range(m)],
            "ttms": t.ctypes.data_as(dp), "forwards": f.ctypes.data_as(dp), "discfactors": d.ctypes.data_as(dp),
            "strikes": k_all.ctypes.data_as(dp), "codes": c_all.ctypes.data_as(C.POINTER(C.c_int8)),
            "offsets": offs.ctypes.data_as(C.POINTER(C.c_size_t)), "nbs": (C.c_int * m)(*self.nb_steps),
            "dts": dts.ctypes.data_as(dp),
        }
        if len(cache) >= 8:
            cache.clear()
        cache[key] = hit
        return hit

    def price_logsv_chain_sets(self, ttms, forwards, discfactors, strikes: Sequence[np.ndarray], codes: Sequence[np.ndarray],
                               params_rows: np.ndarray, is_spot_measure: bool, variable_type: int, want_ivols: bool = False,
                               use_graph: bool = True, comm_handle=None, rank: int = 0, world: int = 1):
        """SEVERAL parameter sets on these randoms in one call of svmc_logsv_chain_price_fixed_sets: with 2..8 sets one
        replayed graph whose stepping launch reads the randoms once for all of them.  params_rows [n_sets][6 + m] =
        (v0, theta, kappa1, kappa2, beta, volvol, vol-backbone eta per expiry).  Returns per set what price_logsv_chain
        returns -- the same bits."""
        lib = _lib.load()
        m = len(self)
        params_rows = np.ascontiguousarray(params_rows, dtype=np.float64)
        n_sets = params_rows.shape[0]
        if params_rows.ndim != 2 or params_rows.shape[1] != 6 + m or n_sets < 1:
            raise ValueError("params_rows must have shape [n_sets, 6 + n_expiries]")
        ch = self._marshalled_chain(ttms, forwards, discfactors, strikes, codes)
        offs, total = ch["offs"], ch["total"]
        per_launch = min(n_sets, 8)
        need = (m * per_launch, max(total * per_launch, 1))
        if self.is_frozen:
            need = (m * 8, max(total * 8, 1))       # one session for every set count of an optimizer run (1 .. 8 per launch)
        if self._session_sets is None or self._session_sets_size[0] < need[0] or self._session_sets_size[1] < need[1]:
            if self._session_sets is not None:
                _lib.check(lib.svmc_session_destroy(self._session_sets))
            sess = C.c_void_p()
            _lib.check(lib.svmc_session_create(C.byref(sess), self.n_local, need[0], need[1]))
            self._session_sets, self._session_sets_size = sess, need
            self._graphs_on = None
            if comm_handle is not None:
                _lib.check(lib.svmc_session_set_comm(sess, comm_handle, int(rank), int(world), self.nb_path, self.col0))
        dp = C.POINTER(C.c_double)
        # one result block per call, cut into per-set, per-expiry VIEWS (nothing else refers to the block)
        res = np.empty((3 if want_ivols else 2, n_sets, max(total, 1)))
        p_res = res.ctypes.data
        row_bytes = res.strides[0]
        ptr = lambda j: C.cast(p_res + j * row_bytes, dp)      # noqa: E731
        slices = ch["slices"]
        if self.is_frozen:
            if self.__dict__.get("_graphs_on") != bool(use_graph):
                _lib.check(lib.svmc_session_use_graphs(self._session_sets, int(bool(use_graph))))
                self._graphs_on = bool(use_graph)
            seed, call_id = self.frozen_stream
            _lib.check(lib.svmc_logsv_chain_price_frozen_sets(
                self._session_sets, ch["ttms"], ch["forwards"], ch["discfactors"], m, ch["strikes"], ch["codes"], ch["offsets"],
                n_sets, params_rows.ctypes.data_as(dp), int(bool(is_spot_measure)), int(variable_type), ch["nbs"], ch["dts"],
                seed, call_id, ptr(0), ptr(1), ptr(2) if want_ivols else None))
        else:
            w0 = (C.c_void_p * m)(*[b.ptr for b in self.w0])
            w1 = (C.c_void_p * m)(*[b.ptr for b in self.w1])
            _lib.check(lib.svmc_logsv_chain_price_fixed_sets(
                self._session_sets, ch["ttms"], ch["forwards"], ch["discfactors"], m, ch["strikes"], ch["codes"], ch["offsets"],
                n_sets, params_rows.ctypes.data_as(dp), int(bool(is_spot_measure)), int(variable_type), w0, w1, ch["nbs"],
                ch["dts"], self.n_local, ptr(0), ptr(1), ptr(2) if want_ivols else None))
        return [tuple([part[q, sl] for sl in slices] for part in res) for q in range(n_sets)]


def payoff_finalize(sums: np.ndarray, shifts: np.ndarray, discfactor: float, n_path_total: float
                    ) -> Tuple[np.ndarray, np.ndarray]:
    """host arithmetic of utils/mc_payoffs.py:85-88 on the reduced sums (svmc_payoff_finalize)."""
    lib = _lib.load()
    k = len(shifts)
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    shifts = np.ascontiguousarray(shifts, dtype=np.float64)
    prices, stderrs = np.empty(k), np.empty(k)
    pd = C.POINTER(C.c_double)
    _lib.check(lib.svmc_payoff_finalize(sums.ctypes.data_as(pd), shifts.ctypes.data_as(pd), k, float(discfactor),
                                        float(n_path_total), prices.ctypes.data_as(pd), stderrs.ctypes.data_as(pd)))
    return prices, stderrs


def payoff_finalize_chain(sums: np.ndarray, shifts: np.ndarray, discfactors: np.ndarray, n_path_total: float
                          ) -> Tuple[np.ndarray, np.ndarray]:
    """payoff_finalize for all the strikes of a chain in one call (svmc_payoff_finalize_chain): discfactors holds one
    discount factor per strike.  The same arithmetic per strike, hence the same bits as expiry-by-expiry calls."""
    lib = _lib.load()
    k = len(shifts)
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    shifts = np.ascontiguousarray(shifts, dtype=np.float64)
    discfactors = np.ascontiguousarray(discfactors, dtype=np.float64)
    prices, stderrs = np.empty(k), np.empty(k)
    pd = C.POINTER(C.c_double)
    _lib.check(lib.svmc_payoff_finalize_chain(sums.ctypes.data_as(pd), shifts.ctypes.data_as(pd), discfactors.ctypes.data_as(pd),
                                              k, float(n_path_total), prices.ctypes.data_as(pd), stderrs.ctypes.data_as(pd)))
    return prices, stderrs


# Engines are cached per (device id, n_path, path_offset, THREAD) so that buffers stay resident across calls and two threads
# that price chains of the same size concurrently never share state buffers, reduction scratch or the pinned download
# buffer: each gets its own engine (the launches of all of them are serialised by the HIP runtime on the stream they
# were given -- the default stream unless the caller built its own HipEngine with another).  The cache is process-global and
# guarded by a lock.  Eviction (more than MAX_CACHED_ENGINES resident) only ever closes an engine nobody else references
# -- a caller that kept the object returned by get_engine() keeps its HBM; the engines of threads that ended are the
# first to go -- and an engine that was closed raises SvmcError (null pointer) on its next launch instead of touching
# freed memory.
MAX_CACHED_ENGINES = 4
_ENGINES = {}
_ENGINES_LOCK = threading.Lock()


def _current_device() -> int:
    dev = C.c_int()
    _lib.check(_lib.load().svmc_get_device(C.byref(dev)))
    return dev.value


def get_engine(n_path: int, path_offset: int = 0, device: Optional[int] = None) -> HipEngine:
    dev = _current_device() if device is None else int(device)     # None = the CURRENT HIP device, resolved now
    key = (dev, int(n_path), int(path_offset), threading.get_ident())
    with _ENGINES_LOCK:
        eng = _ENGINES.get(key)
        if eng is not None and not eng.closed:
            _ENGINES[key] = _ENGINES.pop(key)          # most recently used last
            return eng
        if len(_ENGINES) >= MAX_CACHED_ENGINES:        # bound resident HBM: drop FREE engines, dead threads' first, then LRU
            alive = {t.ident for t in threading.enumerate()}
            for k in sorted(_ENGINES, key=lambda k_: k_[3] in alive):      # stable: keeps the LRU order within each class
                # references: the dict, the loop variable below, getrefcount's argument
                cand = _ENGINES[k]
                if sys.getrefcount(cand) <= 3:
                    _ENGINES.pop(k).close()
                    break
                del cand
        eng = HipEngine(n_path, device=dev, path_offset=path_offset)
        _ENGINES[key] = eng
        return eng
