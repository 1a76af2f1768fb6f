"""Device-resident ensembles and multi-GPU sharding.

PyTorch is plumbing only (device memory, streams, ``torch.distributed`` = RCCL on ROCm): tensors are
handed to the C ABI as raw device pointers.  One process per GPU; chains are independent, so ranks
take contiguous chain ranges, draw from streams keyed by the GLOBAL chain id (results
do not depend on the number of GPUs) and exchange nothing while sampling.  The only collective is the
optional final all-gather of skeletons / samples, or an all-reduce of moment sums.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ArgumentError, DimensionMismatch
from .samplers import AbstractPDMP
from .sample import _seed


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


def _on_stream(dev, stream):
    """Context of a device entry point: ``dev`` current and -- when the caller passed a stream -- that stream torch's current one, so
    that the torch-side pieces of a host loop (zero fills, clones, ``.item()`` reads) are ordered with the engine's kernels, which are
    enqueued on ``stream``."""
    import contextlib
    st = contextlib.ExitStack()
    st.enter_context(torch.cuda.device(dev))
    if stream is not None:
        st.enter_context(torch.cuda.stream(stream))
    return st


def _need_cuda(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype not in (torch.float64, torch.int32, torch.int64, torch.uint8) or not t.is_contiguous()):
            raise ArgumentError("device entry points need contiguous float64 / int32 / int64 / uint8 CUDA tensors "
                                "(the engine has no CPU fallback)")


class _ChunkedBuffer:
    """Device memory from pdmp_device_alloc, exposed through ``__cuda_array_interface__`` so that torch wraps it without a copy
    (``torch.as_tensor`` keeps this object alive for as long as a tensor views the memory; the memory is released -- after a device
    synchronisation -- when the last view is gone)."""

    def __init__(self, nbytes: int, device: int, chunk_bytes: int = 0):
        p = C.c_void_p()
        rc = _lib.lib().pdmp_device_alloc(int(device), int(nbytes), int(chunk_bytes), C.byref(p))
        if rc != 0:
            raise _lib.EngineError(rc, (_lib.lib().pdmp_device_alloc_error() or b"").decode())
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                _lib.lib().pdmp_device_free(C.c_void_p(ptr))
            except Exception:                                    # interpreter shutdown
                pass


CHUNKED_FROM_BYTES = 1 << 30          # output arrays from this size on are built from physical chunks by default


def alloc_chunked(shape, dtype=torch.float64, device=None, chunk_bytes: int = 0) -> torch.Tensor:
    """An uninitialised tensor whose memory is built from independent 1-GiB physical chunks (pdmp_device_alloc):
    for the large output arrays of an ensemble.  hipMalloc backs a 40 GB array with a few huge physical blocks, and the launch that
    streams X and V side by side then runs 14.4 ms or ~18 ms depending on how those blocks lie relative to each other in the memory's
    channel map, for as long as the allocation lives; chunked arrays run 14.4 - 14.75 ms in every allocation measured."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = 1
    for k in shape:
        n *= int(k)
    esize = torch.empty((), dtype=dtype).element_size()
    buf = _ChunkedBuffer(max(n * esize, 1), dev.index if dev.index is not None else torch.cuda.current_device(), chunk_bytes)
    with torch.cuda.device(dev):
        flat = torch.as_tensor(buf, device=dev)                  # uint8 view of the mapping, zero-copy; holds `buf`
    return flat[: n * esize].view(dtype).reshape(tuple(int(k) for k in shape))


class _StreamBuffer(_ChunkedBuffer):
    """One array of pdmp_device_alloc_streams (same ownership rules as _ChunkedBuffer: released with pdmp_device_free)."""

    def __init__(self, ptr: int, nbytes: int):
        self.ptr, self.nbytes = int(ptr), int(nbytes)
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}


def pool_trim() -> None:
    """Release the device memory parked in the allocator's pool (pdmp_device_pool_trim): freed arrays of pdmp_device_alloc /
    pdmp_device_alloc_streams wait there, mapping intact, for the next request of the same size."""
    rc = _lib.lib().pdmp_device_pool_trim()
    if rc != 0:
        raise _lib.EngineError(rc, (_lib.lib().pdmp_device_alloc_error() or b"").decode())


DEFAULT_DRAWS = 3                      # sets of (X, V, t) drawn by alloc_outputs for arrays of at least CHUNKED_FROM_BYTES
last_alloc_info: dict = {}             # what the last alloc_outputs call measured: {"draws": k, "draw_ms": [...], "seconds": s}


def alloc_outputs(C_: int, n_sk: int, d: int, device=None, chunked: bool | None = None, draws: int | None = None) -> dict:
    """The output arrays of ``sample_skeleton_device`` for ``C`` chains: ``X, V (C, n_sk, d)``, ``t (C, n_sk)``.  ``chunked`` = None:
    arrays of at least 1 GiB from physical chunks, smaller ones from torch's allocator.  Chunked sets are DRAWN up to ``draws`` times
    (default 3, as far as the free memory holds a second set) and the set on which a kernel with the launch's store pattern runs fastest
    is kept (pdmp_device_alloc_streams; DESIGN.md section 6b: the same launch takes 14.3 ms or 17 - 19 ms depending on where the physical
    chunks of X and V lie relative to each other, and a fresh set is a fresh draw).  ``last_alloc_info`` keeps the times measured."""
    import time
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    shapes = (("X", (C_, n_sk, d)), ("V", (C_, n_sk, d)), ("t", (C_, n_sk)))
    nbytes = [8 * C_ * n_sk * d, 8 * C_ * n_sk * d, 8 * C_ * n_sk]
    use = chunked if chunked is not None else nbytes[0] >= CHUNKED_FROM_BYTES
    last_alloc_info.clear()
    if not use or min(nbytes) <= 0:
        return {name: torch.empty(shape, dtype=torch.float64, device=dev) for name, shape in shapes}
    k = DEFAULT_DRAWS if draws is None else max(1, int(draws))
    sizes = (C.c_size_t * 3)(*nbytes)
    ptrs = (C.c_void_p * 3)()
    ms = (C.c_float * k)()
    drawn = C.c_int32(0)
    t0 = time.perf_counter()
    with torch.cuda.device(dev):
        rc = _lib.lib().pdmp_device_alloc_streams(dev.index if dev.index is not None else torch.cuda.current_device(), 3, sizes, 0, 8 * n_sk * d, k,
                                                  ptrs, ms, C.byref(drawn))
    if rc != 0:
        raise _lib.EngineError(rc, (_lib.lib().pdmp_device_alloc_error() or b"").decode())
    last_alloc_info.update({"draws": int(drawn.value), "draw_ms": [round(float(ms[i]), 3) for i in range(int(drawn.value))] if k > 1 else [],
                            "seconds": round(time.perf_counter() - t0, 3)})
    out = {}
    with torch.cuda.device(dev):
        for (name, shape), ptr, nb in zip(shapes, ptrs, nbytes):
            flat = torch.as_tensor(_StreamBuffer(ptr, nb), device=dev)       # zero-copy view; holds the buffer object, which frees the memory
            out[name] = flat.view(torch.float64).reshape(shape)
    return out


V_RADEMACHER, V_GAUSSIAN, V_SPHERE = 0, 1, 2


def init_states_device(sampler: AbstractPDMP, C_: int, *, seed, chain_id_offset: int = 0, x_mean: float = 0.0, x_scale: float = 1.0,
                       v_kind: int = V_RADEMACHER, device=None, stream=None):
    """pdmp_init_states_device: synthetic initial states of chains ``chain_id_offset .. +C`` generated on the device from stream 1 of
    each chain's Philox key -- ``x = x_mean + x_scale N(0, I)``; ``v``: +-1 coordinates, ``N(0, I)`` or uniform
    on the unit sphere.  Keyed by the GLOBAL chain id: a shard's states do not depend on the sharding.  Returns ``(x, v)``, ``(C, d)``."""
    if C_ <= 0:
        raise ArgumentError(f"C must be positive. Current value: {C_}")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    x = torch.empty((C_, sampler.dim), dtype=torch.float64, device=dev)
    v = torch.empty((C_, sampler.dim), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pdmp_init_states_device(sampler.handle(), int(C_), int(chain_id_offset), C.c_uint64(_seed(seed)), float(x_mean),
                                                      float(x_scale), int(v_kind), _ptr(x), _ptr(v), _stream_ptr(stream)))
    return x, v


def sample_skeleton_device(sampler: AbstractPDMP, n_sk: int, xinit: torch.Tensor, vinit: torch.Tensor, *,
                           seed=None, chain_id_offset: int = 0, diagnostics: bool = False, out: dict | None = None,
                           stream=None, check: bool = True) -> dict:
    """pdmp_sample_skeleton_device on HBM-resident ``(C, d)`` inputs; outputs stay on the device:
    X, V ``(C, n_sk, d)``, t ``(C, n_sk)`` (+ diagnostics).  ``out`` may supply preallocated tensors."""
    if n_sk <= 0:
        raise ArgumentError(f"n_sk must be positive. Current value: {n_sk}")
    if xinit.dim() != 2 or xinit.shape != vinit.shape or xinit.shape[1] != sampler.dim:
        raise DimensionMismatch(f"xinit and vinit must be (C, {sampler.dim})")
    _need_cuda(xinit, vinit)
    Cn, d = xinit.shape
    dev = xinit.device
    o = {} if out is None else out
    def buf(name, shape, dtype=torch.float64):
        tns = o.get(name)
        if tns is None or tuple(tns.shape) != tuple(shape):
            nbytes = torch.empty((), dtype=dtype).element_size()
            for k in shape:
                nbytes *= int(k)
            # large arrays from independent physical chunks: their launch time does not depend on the placement lottery of hipMalloc --
            # in a single-process run only: memory mapped through HIP's virtual-memory API is enabled for the local device alone and has
            # never been handed to RCCL on hardware, so a multi-rank run keeps torch's allocator for everything it may exchange
            import torch.distributed as _dist
            multi = _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1
            tns = alloc_chunked(shape, dtype, dev) if (nbytes >= CHUNKED_FROM_BYTES and not multi) else torch.empty(shape, dtype=dtype, device=dev)
            o[name] = tns
        _need_cuda(tns)
        return tns
    X, V, t = buf("X", (Cn, n_sk, d)), buf("V", (Cn, n_sk, d)), buf("t", (Cn, n_sk))
    diag = None
    if diagnostics:
        diag = _lib.Diag(buf("horizon", (Cn, n_sk)).data_ptr(), buf("ar", (Cn, n_sk)).data_ptr(),
                         buf("errored_bound", (Cn, n_sk), torch.int32).data_ptr(),
                         buf("error_value_ar", (Cn, n_sk, 5)).data_ptr(),
                         buf("rejected", (Cn, n_sk), torch.int32).data_ptr(),
                         buf("hitting_horizon", (Cn, n_sk), torch.int32).data_ptr(), None)
    if sampler.sampler_id == 6:                              # StickyZigZag: PDMPHistory.is_active (Composites.jl:142,256)
        if diag is None:
            diag = _lib.Diag(None, None, None, None, None, None, None)
        diag.is_active = buf("is_active", (Cn, n_sk, d), torch.uint8).data_ptr()
    L = _lib.lib()
    with torch.cuda.device(dev):
        _lib.check(L.pdmp_sample_skeleton_device(sampler.handle(), Cn, int(n_sk), int(chain_id_offset), _ptr(xinit),
                                                 _ptr(vinit), C.c_uint64(_seed(seed)), _ptr(X), _ptr(V), _ptr(t),
                                                 C.byref(diag) if diag is not None else None, _stream_ptr(stream)))
        if check:
            _lib.check(L.pdmp_check_status(sampler.handle(), None))
    return o


def _diag_of(out: dict):
    """pdmp_diag_t over whichever diagnostic tensors ``out`` holds (None when it holds none)."""
    keys = ("horizon", "ar", "errored_bound", "error_value_ar", "rejected", "hitting_horizon", "is_active")
    if not any(k in out for k in keys):
        return None
    return _lib.Diag(*[out[k].data_ptr() if k in out else None for k in keys])


def sample_from_skeleton_device(sampler: AbstractPDMP, N: int, X: torch.Tensor, V: torch.Tensor, t: torch.Tensor, *,
                                discard_vt: bool = True, out: torch.Tensor | None = None, stream=None,
                                is_active: torch.Tensor | None = None, lens: torch.Tensor | None = None):
    """pdmp_sample_from_skeleton_device: returns ``(C, N, rows)`` on the device.  ``is_active`` (uint8, like X): required for
    StickyZigZag (sample.jl:516-561); ``lens`` (int64, (C,)): ragged history lengths of the time-horizon variant."""
    if N <= 0:
        raise ArgumentError(f"N must be positive. Current value: {N}")
    _need_cuda(X, V, t, is_active, lens)
    Cn, n_sk, d = X.shape
    if d != sampler.dim or V.shape != X.shape or tuple(t.shape) != (Cn, n_sk):
        raise DimensionMismatch("history tensors do not match the sampler dimension")
    rows = d if discard_vt else 2 * d + 1
    if out is None:
        out = torch.empty((Cn, N, rows), dtype=torch.float64, device=X.device)
    _need_cuda(out)
    with torch.cuda.device(X.device):
        _lib.check(_lib.lib().pdmp_sample_from_skeleton_device(sampler.handle(), Cn, n_sk, _ptr(X), _ptr(V), _ptr(t),
                                                               _ptr(is_active), _ptr(lens), int(N), int(bool(discard_vt)),
                                                               _ptr(out), _stream_ptr(stream)))
    return out


def sample_moments_device(sampler: AbstractPDMP, N: int, X: torch.Tensor, V: torch.Tensor, t: torch.Tensor, *, stream=None,
                          is_active: torch.Tensor | None = None, lens: torch.Tensor | None = None):
    """pdmp_sample_moments_device: per-chain sums of x and x^2 over the N samples sample_from_skeleton would
    return (never materialised).  Returns (sum_x, sum_x2), each ``(C, d)`` on the device."""
    if N <= 0:
        raise ArgumentError(f"N must be positive. Current value: {N}")
    _need_cuda(X, V, t, is_active, lens)
    Cn, n_sk, d = X.shape
    if d != sampler.dim or V.shape != X.shape or tuple(t.shape) != (Cn, n_sk):
        raise DimensionMismatch("history tensors do not match the sampler dimension")
    s1 = torch.empty((Cn, d), dtype=torch.float64, device=X.device)
    s2 = torch.empty((Cn, d), dtype=torch.float64, device=X.device)
    with torch.cuda.device(X.device):
        _lib.check(_lib.lib().pdmp_sample_moments_device(sampler.handle(), Cn, n_sk, _ptr(X), _ptr(V), _ptr(t), _ptr(is_active),
                                                         _ptr(lens), int(N), _ptr(s1), _ptr(s2), _stream_ptr(stream)))
    return s1, s2


def compact_history_device(sampler: AbstractPDMP, out: dict, count: torch.Tensor, *, stream=None) -> dict:
    """``_trim_history`` (Composites.jl:198-214) for an ensemble on the device: the ragged histories ``[0, count[c])`` of ``out``
    (tensors shaped ``(C, n_sk, ...)``) packed back to back.  Returns ``offsets`` ``(C+1,)`` int64 (wave prefix scan,
    pdmp_compact_offsets_device) and dense tensors ``X, V (total, d)``, ``t (total,)`` + the diagnostics ``out`` holds."""
    _need_cuda(count)
    Cn, n_sk, d = out["X"].shape
    dev = out["X"].device
    L = _lib.lib()
    with _on_stream(dev, stream):
        offsets = torch.empty((Cn + 1,), dtype=torch.int64, device=dev)
        _lib.check(L.pdmp_compact_offsets_device(sampler.handle(), Cn, _ptr(count), _ptr(offsets), _stream_ptr(stream)))
        total = int(offsets[-1].item())           # (.item() waits on torch's current stream = `stream` inside this context)
        res = {"offsets": offsets, "count": count,
               "X": torch.empty((total, d), dtype=torch.float64, device=dev), "V": torch.empty((total, d), dtype=torch.float64, device=dev),
               "t": torch.empty((total,), dtype=torch.float64, device=dev)}
        for k, shape in (("horizon", (total,)), ("ar", (total,)), ("errored_bound", (total,)), ("error_value_ar", (total, 5)),
                         ("rejected", (total,)), ("hitting_horizon", (total,)), ("is_active", (total, d))):
            if k in out:
                res[k] = torch.empty(shape, dtype=out[k].dtype, device=dev)
        din, dout = _diag_of(out), _diag_of(res)
        _lib.check(L.pdmp_compact_history_device(sampler.handle(), Cn, n_sk, _ptr(count), _ptr(offsets), _ptr(out["X"]), _ptr(out["V"]),
                                                 _ptr(out["t"]), C.byref(din) if din is not None else None, _ptr(res["X"]),
                                                 _ptr(res["V"]), _ptr(res["t"]), C.byref(dout) if dout is not None else None,
                                                 _stream_ptr(stream)))
    return res


_DIAG_SPECS = (("horizon", torch.float64, ()), ("ar", torch.float64, ()), ("errored_bound", torch.int32, ()),
               ("error_value_ar", torch.float64, (5,)), ("rejected", torch.int32, ()), ("hitting_horizon", torch.int32, ()))


def sample_skeleton_until_chunked_device(sampler: AbstractPDMP, T: float, xinit: torch.Tensor, vinit: torch.Tensor, *, seed=None,
                                         chain_id_offset: int = 0, diagnostics: bool = False, init_capacity: int = 1024,
                                         stream=None, stats: dict | None = None, compact: bool = True) -> dict:
    """Time-horizon variant ``sample_skeleton(sampler, T::Float64, xinit, vinit; seed, init_capacity)`` (src/sample.jl:323-439) for an
    ensemble, with the time horizon as a KERNEL feature (include/pdmp_hip.h, ABI 8): the history is produced in chunks -- one launch
    over the chains that have not reached ``T`` yet; a chain stops inside the kernel after its first event with ``t >= T``
    (``while state.t < T``, :372), the others use the chunk's whole capacity and resume in the next chunk from their hand-off
    state; finished chains cost nothing.  Between chunks, on the device: the closing column at exactly ``t == T`` (:385-420,
    ``pdmp_close_chunk_device``), the per-chain history lengths and the compacted list of the chains still running (ballot +
    prefix sums, ``pdmp_until_advance_device``); the host learns ONE number per chunk, how many chains are left.  Memory is the
    sum of the chunks -- O(sum of the history lengths) -- plus the dense result, never chains x longest history.

    Chunk capacities: the first is ``init_capacity`` (the reference's keyword; rounded up to a multiple of 8 so that a chunk's rows
    start on 64-byte lines); the reference then doubles its one chain's buffer (:375-377, Composites.jl:172-191) -- here a later
    chunk is sized from what the chains still running have left: the 90 % quantile of (T - t) x their event rate so far, with 10 % + 8
    columns of slack (at most four times the capacity so far); stragglers take one more, small chunk.

    Returns the histories trimmed and packed back to back (``_trim_history``, Composites.jl:198-214): ``X, V (total, d)``, ``t
    (total,)`` (+ diagnostics), ``offsets (C+1,)``, ``count (C,)``: chain c = rows ``offsets[c] : offsets[c+1]``.
    ``compact=False`` skips that copy (it reads and writes every event once more -- at config-2 size twice the time of the
    sampling itself): when ONE chunk held every chain (a first capacity above the longest history) the result is that chunk as it
    is, ``X, V (C, capacity, d)``, ``t (C, capacity)`` + ``count`` -- the ragged layout ``sample_from_skeleton_device(lens=count)``
    and the fused moments read directly; otherwise ``{"chunks": [...], "count": ...}``: per chunk its tensors, ``count`` (columns;
    negative: -n-1 for a chain that continues), ``start`` (where they go in the chain's history), ``map`` (launch chain ->
    chain, None: identity) and ``skip`` (1: column 0 repeats the previous chunk's last event).
    Raises EngineError(UNSUPPORTED) for configurations the in-kernel horizon does not serve (dense precision, the wave-per-chain
    kernel, tape replay): ``sample_skeleton_until_device`` falls back to the capacity-doubling re-run for those."""
    import math
    T = float(T)
    if not math.isfinite(T) or T < 0:
        raise ArgumentError(f"T must be finite and non-negative. Current value: {T}")      # sample.jl:335-337
    if xinit.dim() != 2 or xinit.shape != vinit.shape or xinit.shape[1] != sampler.dim:
        raise DimensionMismatch(f"xinit and vinit must be (C, {sampler.dim})")
    _need_cuda(xinit, vinit)
    Cn, d = xinit.shape
    dev = xinit.device
    seed = _seed(seed)
    L = _lib.lib()
    h = sampler.handle()
    sp = _stream_ptr(stream)
    # the lane forms address a wave's window of X / V (64 chains x capacity x d doubles) through a 32-bit buffer descriptor
    # (engine_api.hip: lane_ok) and a resumed chunk must run in the first chunk's form: keep every capacity below that window
    cap_limit = ((2 ** 31 - 1) // (64 * 8 * d)) // 8 * 8 if d <= 104 else 1 << 40
    cap = min(-(-max(2, int(init_capacity)) // 8) * 8, max(cap_limit, 8))
    chunks, cmap, C_l, cap_sum, allocated = [], None, Cn, 0, 0
    prof = [] if (stats is not None and stats.get("profile")) else None

    def mark(what):                                  # diagnostic: wall time of each step, synchronised (stats={"profile": True})
        if prof is not None:
            import time
            torch.cuda.synchronize()
            prof.append((what, time.perf_counter()))
    with _on_stream(dev, stream):
        total = torch.zeros((Cn,), dtype=torch.int64, device=dev)
        while True:
            mark("chunk begin")
            bufs = {"X": torch.empty((C_l, cap, d), dtype=torch.float64, device=dev), "V": torch.empty((C_l, cap, d), dtype=torch.float64, device=dev),
                    "t": torch.empty((C_l, cap), dtype=torch.float64, device=dev)}
            if diagnostics:
                for k, dt_, tail in _DIAG_SPECS:
                    bufs[k] = torch.empty((C_l, cap) + tail, dtype=dt_, device=dev)
            if sampler.sampler_id == 6:                          # StickyZigZag: PDMPHistory.is_active (Composites.jl:142,256)
                bufs["is_active"] = torch.empty((C_l, cap, d), dtype=torch.uint8, device=dev)
            allocated += sum(b.numel() * b.element_size() for b in bufs.values())
            dg = _diag_of(bufs)
            nrec = torch.empty((C_l,), dtype=torch.int64, device=dev)
            count = torch.empty((C_l,), dtype=torch.int64, device=dev)
            start = torch.empty((C_l,), dtype=torch.int64, device=dev)
            nxt = torch.empty((C_l,), dtype=torch.int64, device=dev)
            n_act = torch.zeros((1,), dtype=torch.int64, device=dev)
            resume = 0 if cmap is None else 1
            mark("allocated")
            _lib.check(L.pdmp_sample_skeleton_until_device(h, C_l, cap, int(chain_id_offset), _ptr(xinit), _ptr(vinit), C.c_uint64(seed),
                                                           C.c_double(T), resume, _ptr(cmap), _ptr(bufs["X"]), _ptr(bufs["V"]), _ptr(bufs["t"]),
                                                           C.byref(dg) if dg is not None else None, _ptr(nrec), sp))
            mark("sampled")
            _lib.check(L.pdmp_close_chunk_device(h, C_l, cap, _ptr(bufs["X"]), _ptr(bufs["V"]), _ptr(bufs["t"]),
                                                 C.byref(dg) if dg is not None else None, C.c_double(T), _ptr(nrec), _ptr(count), sp))
            mark("closed")
            _lib.check(L.pdmp_until_advance_device(h, C_l, _ptr(count), resume, _ptr(cmap), _ptr(total), _ptr(start), _ptr(nxt), _ptr(n_act), sp))
            mark("advanced")
            _lib.check(L.pdmp_check_status(h, None))                           # (waits for the chunk; iteration cap / lost hand-off -> error)
            chunks.append({"bufs": bufs, "count": count, "start": start, "map": cmap, "cap": cap, "skip": resume, "C": C_l})
            cap_sum += cap
            n = int(n_act.item())
            if n == 0:
                break
            # the chains still running: history so far, time reached -> columns they still need
            cmap = nxt[:n].clone()
            unfinished = count < 0
            t_reached = bufs["t"][:, cap - 1][unfinished]
            done_cols = total[cmap].to(torch.float64)
            need = (T - t_reached) * (done_cols / t_reached.clamp_min(1e-300))
            # a chunk's capacity is one number for all its chains: sized for the 90 % quantile (+ 10 % + 8 columns) -- the stragglers take
            # one more small chunk, which costs less than giving every chain the capacity of the slowest
            # (below 8192 chains a chunk is small whatever its capacity: sized for the slowest chain, no further chunk)
            if not bool(torch.isfinite(need).all()):
                want = float(cap_sum)
            elif n >= 8192:
                want = float(torch.quantile(need, 0.9).item()) * 1.1 + 8.0
            else:
                want = float(need.max().item()) * 1.25 + 16.0
            cap = int(min(max(16.0, want), float(max(4 * cap_sum, 1024))))   # (an estimate gone wild costs chunks or memory, never correctness)
            cap = min(-(-cap // 8) * 8, max(cap_limit, 8))
            C_l = n
        mark("chunks done")
        if prof is not None:
            stats["profile_ms"] = [(b[0], round((b[1] - a[1]) * 1e3, 3)) for a, b in zip(prof, prof[1:])]
        if stats is not None:
            stats.update(chunks=len(chunks), capacities=[ch["cap"] for ch in chunks], chains=[ch["C"] for ch in chunks], chunk_bytes=allocated)
        if not compact:
            if len(chunks) == 1:
                out = dict(chunks[0]["bufs"])
                out["count"] = total
                return out
            return {"chunks": [{**ch["bufs"], "count": ch["count"], "start": ch["start"], "map": ch["map"], "skip": ch["skip"]} for ch in chunks],
                    "count": total}
        offsets = torch.empty((Cn + 1,), dtype=torch.int64, device=dev)
        _lib.check(L.pdmp_compact_offsets_device(h, Cn, _ptr(total), _ptr(offsets), sp))
        tot = int(offsets[-1].item())
        res = {"offsets": offsets, "count": total, "X": torch.empty((tot, d), dtype=torch.float64, device=dev),
               "V": torch.empty((tot, d), dtype=torch.float64, device=dev), "t": torch.empty((tot,), dtype=torch.float64, device=dev)}
        if diagnostics:
            for k, dt_, tail in _DIAG_SPECS:
                res[k] = torch.empty((tot,) + tail, dtype=dt_, device=dev)
        if sampler.sampler_id == 6:
            res["is_active"] = torch.empty((tot, d), dtype=torch.uint8, device=dev)
        dout = _diag_of(res)
        dense_bytes = sum(res[k].numel() * res[k].element_size() for k in res if k not in ("offsets", "count"))
        for ch in chunks:
            b = ch["bufs"]
            din = _diag_of(b)
            _lib.check(L.pdmp_compact_chunk_device(h, ch["C"], ch["cap"], _ptr(ch["count"]), ch["skip"], _ptr(ch["map"]), _ptr(ch["start"]),
                                                   _ptr(offsets), _ptr(b["X"]), _ptr(b["V"]), _ptr(b["t"]), C.byref(din) if din is not None else None,
                                                   _ptr(res["X"]), _ptr(res["V"]), _ptr(res["t"]), C.byref(dout) if dout is not None else None, sp))
        torch.cuda.current_stream().synchronize() if stream is None else stream.synchronize()   # the chunks are released on return
    if stats is not None:
        stats.update(dense_bytes=dense_bytes)
    return res


def sample_skeleton_until_device(sampler: AbstractPDMP, T: float, xinit: torch.Tensor, vinit: torch.Tensor, *, seed=None,
                                 chain_id_offset: int = 0, diagnostics: bool = False, init_capacity: int = 1024,
                                 stream=None, compact: bool = False) -> dict:
    """Time-horizon variant ``sample_skeleton(sampler, T::Float64, xinit, vinit)`` (src/sample.jl:323-439) for an ensemble
    held on the device: the fixed-count kernel runs with a capacity that doubles (``_grow_history``, Composites.jl:172-191)
    until every chain has passed ``T``; ``pdmp_close_at_time_device`` then writes each chain's closing column at ``t == T``
    in place (:385-420; the sampler's own flow, the Sticky mask :386-392, zeroed statistics :402-408) and returns the ragged
    lengths.  Returns the skeleton tensors plus ``count`` ``(C,)`` int64: chain c's history is columns ``[0, count[c])``;
    columns beyond are the discarded overshoot.  ``compact=True``: the histories trimmed and packed back to back instead
    (``compact_history_device``: dense ``(total, ...)`` tensors + ``offsets``)."""
    import math
    T = float(T)
    if not math.isfinite(T) or T < 0:
        raise ArgumentError(f"T must be finite and non-negative. Current value: {T}")      # sample.jl:335-337
    if compact:                              # packed result: the time horizon inside the kernel, chunked history (memory O(sum of lengths))
        try:
            return sample_skeleton_until_chunked_device(sampler, T, xinit, vinit, seed=seed, chain_id_offset=chain_id_offset,
                                                        diagnostics=diagnostics, init_capacity=init_capacity, stream=stream)
        except _lib.EngineError as e:
            if e.code != _lib.ERR_UNSUPPORTED:
                raise                        # (dense precision, wave-per-chain kernel, tape replay: the re-run below)
    cap = max(2, int(init_capacity))
    seed = _seed(seed)
    while True:
        with _on_stream(xinit.device, stream):               # (the torch-side reduction below must follow the kernels on `stream`)
            out = sample_skeleton_device(sampler, cap, xinit, vinit, seed=seed, chain_id_offset=chain_id_offset,
                                         diagnostics=diagnostics, stream=stream)
            Cn = xinit.shape[0]
            count = torch.empty((Cn,), dtype=torch.int64, device=xinit.device)
            dg = _diag_of(out)
            _lib.check(_lib.lib().pdmp_close_at_time_device(sampler.handle(), Cn, cap, _ptr(out["X"]), _ptr(out["V"]), _ptr(out["t"]),
                                                            C.byref(dg) if dg is not None else None, C.c_double(T),
                                                            C.c_void_p(count.data_ptr()), _stream_ptr(stream)))
            done = bool((count > 0).all())
        if done:
            break
        cap *= 2                                                                          # newcap = 2 * size, :375-377
    if compact:
        return compact_history_device(sampler, out, count, stream=stream)
    out["count"] = count
    return out


def kernel_timing(sampler: AbstractPDMP, reset: bool = False) -> dict:
    """hipEvent-timed kernel durations of this sampler's handle (pdmp_get_timing)."""
    tm = _lib.Timing()
    _lib.check(_lib.lib().pdmp_get_timing(sampler.handle(), C.byref(tm)))
    if reset:
        _lib.check(_lib.lib().pdmp_reset_timing(sampler.handle()))
    return {"kernel_ms_total": tm.kernel_ms_total, "kernel_ms_last": tm.kernel_ms_last, "launches": tm.launches}


# ------------------------------------------------------------------------------------------------
# sharding
# ------------------------------------------------------------------------------------------------
def shard_range(C_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous chain range [lo, hi) of ``rank``: sizes differ by at most one."""
    base, rem = divmod(C_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def batch_shard_range(C_total: int, batch: int, n_batches: int, rank: int, world: int) -> tuple[int, int]:
    """The pipelined partition (``sample_skeleton_sharded(pipeline=B)``): the ensemble is cut into ``n_batches`` contiguous batches,
    each batch into ``world`` contiguous shards; returns the global chain range of ``rank`` inside ``batch``."""
    a, b = shard_range(C_total, batch, n_batches)
    lo, hi = shard_range(b - a, rank, world)
    return a + lo, a + hi


def _all_gather_rows(local: torch.Tensor, C_total: int, world: int, group, via_host: bool = False) -> torch.Tensor:
    """All-gather along dim 0 of per-rank shards of (possibly) unequal length: pad to the longest shard,
    one all_gather_into_tensor (RCCL: one direct exchange per peer link), then drop the padding.
    ``via_host`` (rehearsals of GPU ranks under gloo): through host copies."""
    import torch.distributed as dist
    if via_host and local.is_cuda:
        return _all_gather_rows(local.cpu(), C_total, world, group).to(local.device)
    max_rows = -(-C_total // world)
    pad = max_rows - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    full = local.new_empty((world * max_rows,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(C_total, r, world)
        parts.append(full[r * max_rows: r * max_rows + (hi - lo)])
    return torch.cat(parts, 0) if pad or world > 1 else full


def _gather_in_place(full: torch.Tensor, C_total: int, rank: int, world: int, group, via_host: bool = False, async_op: bool = False) -> list:
    """The all-gather of one output array, in place: ``full`` holds ``C_total`` rows on every rank, rank r has filled rows
    ``shard_range(C_total, r, world)``.  Equal shards: ONE all_gather_into_tensor whose input is this rank's slice of the output
    (RCCL's in-place form: no staging copy; one direct exchange per xGMI peer link).  Unequal shards: one broadcast per shard
    (all-gather-v).  ``via_host`` (rehearsals of GPU ranks under gloo) / CPU tensors: through a copy, gloo does not alias.
    ``async_op`` (device path): the collectives are enqueued behind the work already on the current stream and run on the backend's
    own stream; the returned work handles are waited for by the caller -- kernels launched meanwhile overlap the exchange."""
    import torch.distributed as dist
    works = []
    ranges = [shard_range(C_total, r, world) for r in range(world)]
    lo, hi = ranges[rank]
    equal = len({b - a for a, b in ranges}) == 1
    if via_host or not full.is_cuda:
        mine = full[lo:hi].cpu().clone()
        if equal:
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
        else:
            parts = []
            for r, (a, b) in enumerate(ranges):
                buf = mine if r == rank else torch.empty((b - a,) + tuple(full.shape[1:]), dtype=full.dtype)
                dist.broadcast(buf, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
                parts.append(buf)
        for r, (a, b) in enumerate(ranges):
            if r != rank:
                full[a:b].copy_(parts[r])
        return works
    if equal:
        try:
            w = dist.all_gather_into_tensor(full, full[lo:hi], group=group, async_op=async_op)          # in place: input = this rank's slice of the output
        except (RuntimeError, ValueError):                                        # a backend that refuses aliased buffers (argument check:
            w = dist.all_gather_into_tensor(full, full[lo:hi].clone(), group=group, async_op=async_op)   # every rank takes the same branch)
        if async_op and w is not None:
            works.append(w)
    else:
        for r, (a, b) in enumerate(ranges):
            if b > a:
                w = dist.broadcast(full[a:b], src=dist.get_global_rank(group, r) if group is not None else r, group=group, async_op=async_op)
                if async_op and w is not None:
                    works.append(w)
    return works


def sample_skeleton_sharded(sampler: AbstractPDMP, n_sk: int, xinit: torch.Tensor, vinit: torch.Tensor, *, seed,
                            gather: str = "none", N_samples: int | None = None, group=None, compute=None,
                            out_full: dict | None = None, check: bool = True, via_host: bool = False,
                            lanes: str | int | None = None, pipeline: int = 1) -> dict:
    """Ensemble over all ranks of ``group``.  ``xinit``/``vinit`` hold ALL ``C_total`` chains on every
    rank (cheap: 16 d bytes per chain); this rank samples its contiguous shard.

    gather = "none"      local shard only (keys X, V, t + ``chain_range``)
             "skeleton"  all-gather X, V, t of every chain onto every rank (north-star collective)
             "samples"   run sample_from_skeleton(N_samples) locally and all-gather the (C, N, d) samples
             "moments"   all-reduce sum(x), sum(x^2), count of the local samples (2d+1 doubles)

    ``out_full`` (gather "none" / "skeleton"): preallocated tensors ``X, V (C_total, n_sk, d)``, ``t (C_total, n_sk)`` (for
    "none": shard-sized); the shard is sampled straight into its rows and the gather runs in place (``_gather_in_place``).
    ``lanes``: "auto" picks the lanes-per-chain layout from the largest shard (pdmp_suggest_group_lanes), an int pins it;
    None leaves the handle as it is.  One layout for all ranks -- results are bit-identical across shardings for equal layouts.
    ``pipeline`` = B > 1 (gather "skeleton" with ``out_full``): the exchange overlaps the sampling.  The ensemble is cut into B
    contiguous batches of chains, every rank samples ITS shard of batch j (``batch_shard_range``) and the all-gather of batch j runs
    on the backend's stream while the kernel of batch j + 1 runs on the launch stream: a step costs about max(exchange, sampling)
    instead of their sum.  Chain ids stay global, so every chain's history is bit for bit what the unpipelined run (and a single
    GPU) computes -- only which rank computes it changes.
    ``compute`` (tests only) replaces the HIP engine by a callable
    ``(x_shard, v_shard, chain0) -> {"X","V","t"}`` so the sharding logic can run under gloo on CPU.
    """
    import torch.distributed as dist
    if seed is None:
        raise ArgumentError("a sharded run needs an explicit seed shared by all ranks")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    C_total = xinit.shape[0]
    lo, hi = shard_range(C_total, rank, world)
    if lanes is not None and compute is None:
        G = sampler.suggest_group_lanes(shard_range(C_total, 0, world)[1]) if lanes == "auto" else int(lanes)
        sampler.set_group_lanes(G)
    if pipeline and int(pipeline) > 1:
        if gather != "skeleton" or out_full is None:
            raise ArgumentError("pipeline > 1 overlaps the all-gather of gather = 'skeleton' with the sampling and needs out_full")
        if any(out_full[k].shape[0] != C_total for k in ("X", "V", "t")):
            raise DimensionMismatch(f"out_full tensors must hold {C_total} chains")
        B = min(int(pipeline), max(1, C_total // max(world, 1)))
        works = []
        for j in range(B):
            a, b = shard_range(C_total, j, B)                         # batch j: chains [a, b)
            l2, h2 = batch_shard_range(C_total, j, B, rank, world)    # ... of which this rank samples [l2, h2)
            if h2 > l2:
                piece = {k: out_full[k][l2:h2] for k in ("X", "V", "t")}
                if compute is None:
                    sample_skeleton_device(sampler, n_sk, xinit[l2:h2].contiguous(), vinit[l2:h2].contiguous(), seed=seed, chain_id_offset=l2,
                                           out=piece, check=False)
                else:
                    got = compute(xinit[l2:h2].contiguous(), vinit[l2:h2].contiguous(), l2)
                    for k in ("X", "V", "t"):
                        piece[k].copy_(got[k])
            if world > 1:
                for k in ("X", "V", "t"):
                    works += _gather_in_place(out_full[k][a:b], b - a, rank, world, group, via_host=via_host, async_op=not via_host and out_full[k].is_cuda)
        for w in works:
            w.wait()                                                  # (the launch stream waits for the exchange; the host does not block)
        if check and compute is None:
            _lib.check(_lib.lib().pdmp_check_status(sampler.handle(), None))
        return {"X": out_full["X"], "V": out_full["V"], "t": out_full["t"], "chain_range": (0, C_total), "pipeline": B}
    xs, vs = xinit[lo:hi].contiguous(), vinit[lo:hi].contiguous()
    local_out = None
    if out_full is not None:
        if gather not in ("none", "skeleton", "moments", "samples"):
            raise ArgumentError("out_full serves gather = 'none' / 'skeleton' (and, shard-sized, 'moments' / 'samples')")
        want = C_total if gather == "skeleton" else hi - lo
        if any(out_full[k].shape[0] != want for k in ("X", "V", "t")):
            raise DimensionMismatch(f"out_full tensors must hold {want} chains")
        local_out = {k: (out_full[k][lo:hi] if gather == "skeleton" else out_full[k]) for k in ("X", "V", "t")}
    if compute is None:
        res = sample_skeleton_device(sampler, n_sk, xs, vs, seed=seed, chain_id_offset=lo, out=local_out, check=check)
    else:
        res = compute(xs, vs, lo)
        if local_out is not None:
            for k in ("X", "V", "t"):
                local_out[k].copy_(res[k])
    res = dict(res)
    res["chain_range"] = (lo, hi)
    if gather == "none":
        return res
    if gather == "skeleton":
        if out_full is not None:
            if world > 1:
                for k in ("X", "V", "t"):
                    _gather_in_place(out_full[k], C_total, rank, world, group, via_host=via_host)
            for k in ("X", "V", "t"):
                res[k] = out_full[k]
        elif world > 1:
            for k in ("X", "V", "t"):
                res[k] = _all_gather_rows(res[k], C_total, world, group, via_host=via_host)
        res["chain_range"] = (0, C_total)
        return res
    if N_samples is None or N_samples <= 0:
        raise ArgumentError(f"N must be positive. Current value: {N_samples}")
    if gather == "moments" and compute is None:
        # fused on the device: the d x N x C samples are never written (pdmp_sample_moments_device)
        s1, s2 = sample_moments_device(sampler, N_samples, res["X"], res["V"], res["t"], is_active=res.get("is_active"))
        d = s1.shape[-1]
        m = torch.cat([s1.sum(0), s2.sum(0), s1.new_tensor([float(s1.shape[0] * N_samples)])])
        smp = None
    else:
        smp = (sample_from_skeleton_device(sampler, N_samples, res["X"], res["V"], res["t"], is_active=res.get("is_active"))
               if compute is None else res["samples"])
    if gather == "samples":
        res["samples"] = _all_gather_rows(smp, C_total, world, group, via_host=via_host) if world > 1 else smp
        return res
    if gather == "moments":
        if smp is not None:
            d = smp.shape[-1]
            flat = smp.reshape(-1, d)
            m = torch.cat([flat.sum(0), (flat * flat).sum(0), flat.new_tensor([float(flat.shape[0])])])
        if world > 1:
            if via_host and m.is_cuda:
                mh = m.cpu()
                dist.all_reduce(mh, group=group)
                m = mh.to(m.device)
            else:
                dist.all_reduce(m, group=group)
        n = m[-1]
        mean = m[:d] / n
        res["mean"], res["var"], res["count"] = mean, m[d:2 * d] / n - mean * mean, n
        return res
    raise ArgumentError(f"unknown gather mode {gather!r}")
