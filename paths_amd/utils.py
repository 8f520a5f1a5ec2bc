"""Recursion driver (reference utils.py:228-305), device-resident.

``inference_end2end`` keeps the reference's signature and return values but replaces its per-level host round
trip (``importance.cpu()``, per-slide Python loop with ``torch.topk`` / child expansion / host gather /
``collate_fn``; reference utils.py:248-260, data_utils/slide.py:277-360, data_utils/dataset.py:206-243) by three
kernel launches per level (paths_topk, paths_expand_children, paths_gather_rows).  There is NO host
synchronisation inside the level loop: padded length per level is the static capacity
``4 * min(N_prev, keep)`` instead of the data-dependent per-batch max, padding is masked by ``num_ims``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import contextlib
import torch
import torch.nn.functional as F

from . import _lib, ops
from .data_utils.slide import DeviceSlide, DeviceSlideBatch


def nll_loss(hazards, y, c, alpha=0.4, eps=1e-7):
    """Discrete-time survival NLL (MCAT) — reference utils.py:283-305.  [B,4] tensors: host-side plumbing."""
    B = hazards.shape[0]
    surv = torch.cumprod(1 - hazards, dim=1)
    surv_pad = torch.cat([torch.ones((B, 1), dtype=surv.dtype, device=surv.device), surv], dim=1)
    r = torch.arange(B, device=hazards.device)
    uncensored = -(1 - c) * (torch.log(surv_pad[r, y].clamp(min=eps)) + torch.log(hazards[r, y].clamp(min=eps)))
    censored = -c * torch.log(surv_pad[r, y + 1].clamp(min=eps))
    return ((1 - alpha) * (censored + uncensored) + alpha * uncensored).mean()


class RecursionError_(RuntimeError):
    pass


def recurse(model, slides, keep_patches: Sequence[int], num_levels: int,
            trace: Optional[list] = None, check_status: bool = True) -> Dict[str, torch.Tensor]:
    """Optimistic sync-free pass; if some slide produced zero children (status bit 0, checked once at the end) the
    batch is re-run level by level with the reference's rare fallback (data_utils/slide.py:336-352) handled on the
    device.  See :func:`_recurse` for the arguments."""
    out = _recurse(model, slides, keep_patches, num_levels, trace, careful=False)
    if not check_status:
        return out
    if check_status_word(out["status"]):        # the only host sync of the fast path, after the last level
        if trace is not None:
            trace.clear()
        out = _recurse(model, slides, keep_patches, num_levels, trace, careful=True)
        check_status_word(out["status"], fallback_done=True)
    return out


def check_status_word(status, fallback_done: bool = False) -> bool:
    """Decode the device status word of one recursion (ONE host sync).  Returns True when the batch needs the careful re-run (bit 0:
    some slide produced zero children, reference data_utils/slide.py:336-352); raises on the bits that invalidate the results - bit 1
    (child capacity exceeded) and bit 2 (a bounded in-launch hand-off wait of the token-0 tail gave up, csrc/token0_ws.hip).  Every
    path that returns recursion outputs goes through here: recurse(), GraphedRecursion.run(), TapedRecursion.run() and the
    training forward (autograd.forward_backward)."""
    code = int(status.item()) if torch.is_tensor(status) else int(status)
    if code & 2:
        raise RecursionError_("child capacity exceeded (internal error)")
    if code & 4:
        raise RecursionError_("a bounded in-launch hand-off wait of the token-0 tail gave up (csrc/token0_ws.hip): results invalid")
    return bool(code & 1) and not fallback_done


class GraphedRecursion:
    """One batch's whole recursion (all levels, all three streams) captured ONCE into a HIP graph and replayed per step.

    The optimistic pass is sync-free with static capacities, static slide tables and cached weight images, so a step is a fixed
    launch sequence: capturing it takes Python (14 launches per level, ~1.6 ms of host time per 8-slide step) off the critical
    path - the replay is one hipGraphLaunch (host share of a step 0.65 -> 0.13, bit-identical outputs).  Measured on ROCm 7.2 /
    MI355X the replay runs the three captured streams with much less overlap than eager launches do (3.92 ms against 2.43 ms
    per 8-slide step at K = 2048), so this is an OPTION for host-bound deployments (many ranks per host core), not the default.  The graph owns its buffers (torch's graph-private pool); ``replay()`` returns the same output
    tensors every time (logits, ctx_slide, ctx_patch, importance, status), valid until the next replay.

    Validity: the capture bakes in the weight images that were current at capture time; ``replay()`` re-captures when any
    parameter's version counter has moved (optimizer step, load_state_dict).  The status word (bit 0: a slide without tissue
    children) is NOT handled inside the graph: callers check it like :func:`recurse` does and fall back to ``recurse()`` (eager,
    careful path) for such a batch - :meth:`run` does that."""

    def __init__(self, model, slides, keep_patches: Sequence[int], num_levels: int):
        self.model, self.keep, self.levels = model, list(keep_patches), int(num_levels)
        self.batch = slides if isinstance(slides, DeviceSlideBatch) else DeviceSlideBatch(slides)
        self.graph, self.out, self.versions = None, None, None

    def _param_versions(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def capture(self):
        dev = self.batch.device
        with torch.no_grad():
            # warm-up outside the capture: builds every cached image / table (some of which sync) and the side streams
            _recurse(self.model, self.batch, self.keep, self.levels, None, careful=False)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = _recurse(self.model, self.batch, self.keep, self.levels, None, careful=False)
        self.graph, self.out, self.versions = g, out, self._param_versions()
        return self

    def replay(self) -> Dict[str, torch.Tensor]:
        if self.graph is None or self.versions != self._param_versions():
            self.capture()
        self.graph.replay()
        return self.out

    def run(self) -> Dict[str, torch.Tensor]:
        """replay + the status check of :func:`recurse` (one host sync after the last level)."""
        out = self.replay()
        if check_status_word(out["status"]):
            with torch.no_grad():
                return recurse(self.model, self.batch, self.keep, self.levels)
        return out


class TapedRecursion:
    """One batch's recursion recorded ONCE as a flat launch tape and replayed per step.

    The optimistic pass is sync-free with static capacities, resident slides and cached weight images: a step is a fixed sequence
    of ~70 kernel launches, a dozen cross-stream joins and two zero fills on three HIP streams.  The first call runs it eagerly
    while ``paths_amd._lib`` records every C call with its final arguments (device pointers, sizes, the stream handle); a replay
    is then a tight loop over that list - no tensor allocation, no pack look-ups, no Python-side shape logic - on the SAME three
    streams with the same joins, so the overlap of the eager schedule is kept (a captured HIP graph loses most of it on ROCm 7.2,
    see GraphedRecursion) while the host's share of a step drops from ~0.65 to ~0.15.  Outputs live in the buffers of the recorded
    pass and are overwritten by every replay.

    Validity: the tape holds the weight images current at recording time; ``replay()`` records again when any parameter's version
    counter has moved.  Like the eager fast path it does not handle the zero-children fallback itself: :meth:`run` checks the
    status word and hands such a batch to ``recurse()``.  ``slide_ctx_mode="concat"`` stacks contexts with a torch op per level
    and is not taped (eager path)."""

    def __init__(self, model, slides, keep_patches: Sequence[int], num_levels: int, lane: int = 0):
        self.model, self.keep, self.levels, self.lane = model, list(keep_patches), int(num_levels), int(lane)
        self.batch = slides if isinstance(slides, DeviceSlideBatch) else DeviceSlideBatch(slides)
        self.tape, self.out, self.versions, self.stream_handle = None, None, None, None
        self._unjoined = False               # a replay(join=False) whose streams the caller stream has not waited for yet
        self._rec_batch = None               # private copy of the recorded batch's table TENSORS: what the tape addresses (rebind() re-points them)
        self._events = [[], 0]               # HIP events of the tape's stream joins, re-used when the tape is recorded again
        # The tape holds raw device addresses of every intermediate of the recorded pass.  Those tensors are freed when the pass
        # returns; from the default caching allocator their blocks would be handed to whoever allocates next, and a replay would
        # then write into memory somebody else owns.  So the recorded pass allocates from a pool of its own: freed blocks stay in
        # it (re-used only by this tape's next recording) for as long as the tape lives.
        self._pool = torch.cuda.MemPool()
        if model.procs[0].config.slide_ctx_mode == "concat":
            raise NotImplementedError("TapedRecursion: slide_ctx_mode='concat' is not taped; use recurse()")

    def _param_versions(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def close(self):
        """Drop the tape and destroy the HIP events of its stream joins (a service that records one tape per batch would otherwise
        accumulate about a dozen events per tape).  Called by __del__; the tape can be recorded again afterwards."""
        evs = (getattr(self, "_events", None) or [[], 0])[0]      # (__init__ may have raised before _events existed)
        self._events = [[], 0]
        self.tape = None
        if evs:
            try:
                torch.cuda.synchronize(self.batch.device)       # no replay in flight still waits on them
                lib = _lib.load()
                for ev in evs:
                    lib.paths_event_destroy(ev)
            except Exception:                                    # interpreter shutdown: the runtime may be gone already
                pass

    def __del__(self):
        self.close()

    def record(self):
        global STREAM_LANE
        saved_lane, STREAM_LANE = STREAM_LANE, self.lane
        try:
            with torch.no_grad():
                tab = self.batch.clone_tables()                  # the tape's OWN table tensors (rebind() re-points them)
                _recurse(self.model, tab, self.keep, self.levels, None, careful=False)      # warm-up: builds cached images / tables
                torch.cuda.synchronize(self.batch.device)
                assert _lib.TAPE is None, "a launch tape is already being recorded"
                self.out = None                                   # (a re-recording re-uses the pool's blocks: drop the old outputs first)
                _lib.TAPE = tape = []
                self._events[1] = 0
                _lib.TAPE_EVENTS = self._events
                try:
                    with torch.cuda.use_mem_pool(self._pool, device=self.batch.device):
                        out = _recurse(self.model, tab, self.keep, self.levels, None, careful=False)
                finally:
                    _lib.TAPE, _lib.TAPE_EVENTS = None, None
                torch.cuda.synchronize(self.batch.device)
        finally:
            STREAM_LANE = saved_lane
        # arguments pre-converted to their ctypes types once: a replayed call then skips ctypes' per-argument conversion
        tape = [(fn, tuple(a if a is None else t(a) for t, a in zip(fn.argtypes, args)), name) for fn, args, name in tape]
        self.tape, self.out, self.versions = tape, out, self._param_versions()
        self.stream_handle = torch.cuda.current_stream(self.batch.device).cuda_stream
        self._rec_batch = tab
        return self

    def rebind(self, slides) -> "TapedRecursion":
        """Point the recorded tape at ANOTHER batch of resident slides without recording again (a stream of distinct slides would
        otherwise pay record() per batch or run eager).  The tape addresses the recorded batch's per-level TABLE tensors (grid / mask
        base pointers, grid dims: a private copy made by record()), never the slides themselves, and every capacity in it is static - so a batch
        with the same slide count, feature width and level count whose level-0 cell count, grid extents and operand range fit the
        recorded ones is bound by copying its tables into those tensors (a few hundred bytes per level, device to device, enqueued
        on the current stream: no host sync).  Anything else drops the tape; the next replay records for the new batch."""
        new = slides if isinstance(slides, DeviceSlideBatch) else DeviceSlideBatch(slides)
        rec = self._rec_batch
        fits = (self.tape is not None and rec is not None and len(new) == len(rec) and new.dim == rec.dim and new.device == rec.device
                and new.num_levels >= self.levels and new.n0 <= rec.n0
                and all(new.max_dim[l] <= rec.max_dim[l] for l in range(self.levels))
                and ops.h3_in_range(new.feat_absmax) == ops.h3_in_range(rec.feat_absmax))
        if not fits:
            self.tape, self._rec_batch = None, None
            self.batch = new
            return self
        if self._unjoined:
            # the previous replay(join=False) may still be reading the tables on the lane's streams: the caller stream (which carries
            # the copy below) waits for them first - a write-after-read race otherwise
            self.join()
        with torch.no_grad():
            src, dst = new.flat_tables(), rec.flat_tables()
            if src.numel() == dst.numel():                 # same level count: ONE copy of all tables (a few hundred bytes)
                dst.copy_(src, non_blocking=True)
            else:
                for l in range(self.levels):
                    rec.grid_ptrs[l].copy_(new.grid_ptrs[l], non_blocking=True)
                    rec.mask_ptrs[l].copy_(new.mask_ptrs[l], non_blocking=True)
                    rec.gx[l].copy_(new.gx[l], non_blocking=True)
                    rec.gy[l].copy_(new.gy[l], non_blocking=True)
        self.batch = new                 # (keeps the bound slides alive; run()'s fallback recurses on them)
        return self

    def _play(self, ops_):
        for fn, args, name in ops_:
            if fn(*args) != 0:
                lib = _lib.load()
                msg = lib.paths_last_error().decode()
                lib.paths_clear_stop_event()      # a failure between paths_set_stop_event and the kernel that takes it: leave nothing armed
                raise _lib.PathsHipError(f"{name} failed during tape replay: {msg}")

    def replay(self, join: bool = True) -> Dict[str, torch.Tensor]:
        """join=False leaves out the tape's last entry (the caller stream waiting for this lane's streams): several lanes
        (sub-batches recorded on different stream triples) are then enqueued back to back and run CONCURRENTLY; call
        :meth:`join` on each afterwards."""
        if self.tape is None or self.versions != self._param_versions():
            self.record()
        assert torch.cuda.current_stream(self.batch.device).cuda_stream == self.stream_handle, "replay on the stream the tape was recorded on"
        assert self.tape[-1][2] == "paths_stream_wait"
        self._play(self.tape if join else self.tape[:-1])
        self._unjoined = not join
        return self.out

    def join(self):
        self._play(self.tape[-1:])
        self._unjoined = False

    def run(self) -> Dict[str, torch.Tensor]:
        """replay + the status check of :func:`recurse` (one host sync after the last level)."""
        out = self.replay()
        if check_status_word(out["status"]):
            with torch.no_grad():
                return recurse(self.model, self.batch, self.keep, self.levels)
        return out


class PipelinedRecursion:
    """Several resident batches served with TWO in flight: batch k is recorded as a :class:`TapedRecursion` on stream triple
    ``1 + k % 2`` and :meth:`submit` replays it WITHOUT the tape's final join, so the next batch's selection chain (other streams)
    starts while this batch's last aggregator still runs - the ~165-us gap a single stream triple leaves at every step boundary
    (profiles/r04_experiments.md, selection-queue timeline; bench.py ``host.launch_modes.replay_two_lanes``: +2.7 %).
    :meth:`result` joins batch k into the caller's stream and returns its outputs (valid until that batch is submitted again);
    the status word is checked there like in ``TapedRecursion.run``.  Same launches, same buffers, same results as one lane."""

    def __init__(self, model, batches, keep_patches: Sequence[int], num_levels: int, lanes: int = 2):
        self.tapes = [TapedRecursion(model, b, keep_patches, num_levels, lane=1 + k % max(1, int(lanes))).record() for k, b in enumerate(batches)]

    def submit(self, k: int):
        self.tapes[k].replay(join=False)

    def result(self, k: int, check: bool = True) -> Dict[str, torch.Tensor]:
        t = self.tapes[k]
        t.join()
        if check and check_status_word(t.out["status"]):
            with torch.no_grad():
                return recurse(t.model, t.batch, t.keep, t.levels)
        return t.out

    def close(self):
        for t in self.tapes:
            t.close()
        self.tapes = []


OVERLAP_AGGREGATOR = os.environ.get("PATHS_OVERLAP_AGGREGATOR", "1") != "0"
ZERO_GRAD_ALL = os.environ.get("PATHS_ZERO_GRAD_ALL", "0") != "0"     # train_step: optimizer.zero_grad(set_to_none=True) over ALL parameters (A/B)
ROCTX_RANGES = os.environ.get("PATHS_ROCTX", "0") != "0"     # roctx ranges "level i: selection / aggregator / expansion" around the launches
                                                             # of each level (rocprofv3 --marker-trace; torch.cuda.nvtx = roctx on ROCm)


class _Range:
    """roctx range around a group of launches (host side: marks where the launches were ENQUEUED); a no-op unless PATHS_ROCTX=1."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if ROCTX_RANGES:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if ROCTX_RANGES:
            torch.cuda.nvtx.range_pop()
        return False
ROWS_IN_PLACE = os.environ.get("PATHS_ROWS_IN_PLACE", "1") != "0"
TRAIN_PARENT = os.environ.get("PATHS_TRAIN_PARENT", "1") != "0"      # training: h half of the LSTM gates once per kept parent
_STREAMS: Dict[int, tuple] = {}


STREAM_LANE = 0      # which stream triple _recurse uses (TapedRecursion(lane=k) records each lane on its own streams)


def _streams(dev, lane: Optional[int] = None):
    """(selection-chain stream, aggregator stream, parent-partials stream) of a device.  The selection chain is the critical path of the recursion,
    so it gets the high-priority queue: its workgroups are dispatched first and the aggregator fills what is left."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, STREAM_LANE if lane is None else lane)
    if key not in _STREAMS:
        hi = -1 if os.environ.get("PATHS_STREAM_PRIORITIES", "1") != "0" else 0
        agg = torch.cuda.Stream(device=idx, priority=0)
        ncu = int(os.environ.get("PATHS_AGG_CU_MASK", "0"))
        if ncu > 0:
            # measurement aid: the aggregator stream confined to `ncu` of the chip's compute units, spread evenly over the
            # mask's bit positions (which interleave the XCDs); the selection streams keep the whole chip
            import ctypes
            total = torch.cuda.get_device_properties(idx).multi_processor_count
            words = (total + 31) // 32
            mask = (ctypes.c_uint32 * words)()
            for j in range(ncu):
                bit = (j * total) // ncu
                mask[bit // 32] |= 1 << (bit % 32)
            with torch.cuda.device(idx):
                h = _lib.load().paths_stream_create_masked(mask, words)
            if not h:
                raise _lib.PathsHipError("paths_stream_create_masked failed: " + _lib.load().paths_last_error().decode())
            agg = torch.cuda.ExternalStream(h, device=idx)
        _STREAMS[key] = (torch.cuda.Stream(device=idx, priority=hi), agg, torch.cuda.Stream(device=idx, priority=hi))
    return _STREAMS[key]


def _recurse(model, slides, keep_patches: Sequence[int], num_levels: int,
             trace: Optional[list] = None, careful: bool = False) -> Dict[str, torch.Tensor]:
    batch = slides if isinstance(slides, DeviceSlideBatch) else DeviceSlideBatch(slides)
    with ops.range_guard(batch.feat_absmax):       # out-of-range features run on the exact bf16 split (no fp16 overflow)
        return _recurse_streams(model, batch, keep_patches, num_levels, trace, careful)


def _recurse_streams(model, batch, keep_patches, num_levels, trace, careful):
    if not (OVERLAP_AGGREGATOR and batch.device.type == "cuda"):
        return _recurse_body(model, batch, keep_patches, num_levels, trace, careful, None, None)
    caller = torch.cuda.current_stream(batch.device)
    sel_stream, agg_stream, par_stream = _streams(batch.device)
    if torch.cuda.is_current_stream_capturing():
        # HIP graph capture (GraphedRecursion): on ROCm 7.2 a stream that forks from an already-forked stream crashes
        # hipStreamEndCapture, so every side stream forks from the capture's origin stream: the selection chain runs on the origin
        # stream itself, the aggregator and expansion streams fork from / join into it (the body joins them at its end)
        _lib.stream_wait(agg_stream, caller)
        if par_stream is not None:
            _lib.stream_wait(par_stream, caller)
        return _recurse_body(model, batch, keep_patches, num_levels, trace, careful, agg_stream, par_stream)
    _lib.stream_wait(sel_stream, caller)
    _lib.stream_wait(agg_stream, caller)
    if par_stream is not None:
        _lib.stream_wait(par_stream, caller)
    with torch.cuda.stream(sel_stream):
        out = _recurse_body(model, batch, keep_patches, num_levels, trace, careful, agg_stream, par_stream)
    _lib.stream_wait(caller, sel_stream)     # (the body has already joined agg_stream into sel_stream)
    return out


def _recurse_body(model, slides, keep_patches: Sequence[int], num_levels: int,
                  trace: Optional[list], careful: bool, agg_stream, par_stream=None) -> Dict[str, torch.Tensor]:
    """Run all levels for a batch of HBM-resident slides (a list of DeviceSlide, or a DeviceSlideBatch built once
    and re-used across calls).  Returns the last level's output dict (+ "status").

    ``trace`` (a list) receives one dict per level with device tensors num_ims / locs / parent_inds / importance /
    logits / ctx_slide / keep_idx / keep_count, for parity tests and heat-map export.
    """
    mc = model.procs[0].config
    ops.check_supported(mc)
    batch = slides if isinstance(slides, DeviceSlideBatch) else DeviceSlideBatch(slides)
    assert batch.num_levels >= num_levels
    B = len(batch)
    dev = batch.device
    D = batch.dim
    Dp = model.procs[0].ctx_dim()[1]
    st = _lib.stream()
    p = _lib.ptr
    i32 = dict(device=dev, dtype=torch.int32)
    i64 = dict(device=dev, dtype=torch.int64)
    f32 = dict(device=dev, dtype=torch.float32)

    grid_ptrs, mask_ptrs, gx, gy = batch.grid_ptrs, batch.mask_ptrs, batch.gx, batch.gy
    N = batch.n0
    lstm_pack = ops.pack_lstm(model.lstm) if model.use_lstm else None
    share_parent = model.use_lstm          # siblings share the parent's h: h-half of the gate GEMM once per kept parent
    # default split mode: feature rows are read in place in the resident grids (row-pointer GEMM operands) instead of being
    # copied (level 0) or gathered (children)
    rows_in_place = (share_parent and ops.use_x6(D, Dp - D) and ops.split_planes() == 2 and ROWS_IN_PLACE
                     and (ops.fast_path(mc) or (ops.GENERIC_ADD and ops.GENERIC_SPLIT and D % 128 == 0)))
    zero_row = torch.zeros((D,), **f32) if rows_in_place else None
    fts = None if rows_in_place else torch.empty((B, N, D), **f32)
    x_rows = torch.empty((B, N), **i64) if rows_in_place else None
    locs = torch.empty((B, N, 2), **i64)
    parent_inds = torch.empty((B, N), **i64)
    num_ims = torch.empty((B,), **i64)
    _lib.call("paths_level0_batch", p(grid_ptrs[0]), p(gx[0]), p(gy[0]), B, D, mc.patch_size, N,
              p(fts), p(locs), p(parent_inds), p(num_ims), 0, p(x_rows), p(zero_row), st)
    state_prev, ctx_hist, parent = None, [], None
    out = None
    # The aggregator of level i (attention, token chain, classifier) feeds nothing of level i+1 except the slide context, so
    # it runs on a second HIP stream beside the selection chain of level i+1 (top-K, expansion, gathers, gate GEMMs).  Several
    # of those kernels cannot fill 256 CUs alone (116-232 workgroups, one per CU); the other chain's waves take the idle CUs.
    overlap = agg_stream is not None
    main_stream = torch.cuda.current_stream(dev) if overlap else None
    side_stream = agg_stream
    keepalive = []                         # tensors read on a stream other than the one that allocated them: kept until the streams join
    # importance of padded rows is 0 (reference utils.py:106-115): ONE zero fill for all levels (sizes are known up front
    # unless the careful path has to grow a level)
    sizes, n_l = [], N
    for i in range(num_levels):
        sizes.append(n_l)
        if i < num_levels - 1:
            keep = int(keep_patches[i])
            n_l = 4 * (n_l if keep < 0 else min(n_l, keep))
    # (the status word rides on the same zero fill as the importance rows: one fill launch per step instead of two at its head)
    zbuf = _lib.zeros((B * sum(sizes) + 64,), **f32)
    imp_all = zbuf[:B * sum(sizes)]
    status = zbuf[B * sum(sizes) + 32:B * sum(sizes) + 33].view(torch.int32)
    imp_off = [B * sum(sizes[:i]) for i in range(num_levels)]
    fork_pending = False
    for i in range(num_levels):
        proc = model.procs[i]
        lvl_pack = ops.pack_level(proc)
        if fork_pending:
            _lib.stream_wait(main_stream, par_stream)     # this level's rows / bookkeeping from the expansion branch are ready
            fork_pending = False
        imp_buf = imp_all[imp_off[i]:imp_off[i] + B * N].view(B, N) if N == sizes[i] else None
        # (the aggregator's fork travels as the stop event of the chain's last kernel where that kernel can carry one: _lib.fork_behind)
        # Round 5: where the importance finish also selects the top-K (ops.FUSE_TOPK) the expansion stream's fork rides on the same kernel.
        last = i == num_levels - 1
        keep = None if last else int(keep_patches[i])
        forked = overlap and par_stream is not None and share_parent and not last
        want_topk = forked and rows_in_place and ops.FUSE_TOPK and ops.FUSE_QKV == 2
        dsts = ([side_stream] if overlap else []) + ([par_stream] if want_topk else [])
        with _Range(f"level {i}: selection chain (LSTM gates, importance, projection)"), \
                (_lib.fork_behind(dsts, main_stream) if overlap else contextlib.nullcontext()):
            sel = ops.selection_forward(mc, lstm_pack, lvl_pack, fts, locs, num_ims, state_prev, True, parent=parent,
                                        max_pos=batch.max_dim[i], x_rows=x_rows, feat_dim=D, importance_out=imp_buf,
                                        last_level=last,
                                        topk={"keep": keep, "zero_row": zero_row, "status": status} if want_topk else None)
        fused_topk = "keep_idx" in sel           # (else: the top-K launch below carries the expansion stream's fork, as before)
        def aggregate():
            ctx_prev = ctx_hist[-1] if (ctx_hist and mc.slide_ctx_mode == "residual") else None
            ctx_all = torch.stack(ctx_hist, dim=1) if (ctx_hist and mc.slide_ctx_mode == "concat") else None
            return ops.aggregator_forward(mc, lvl_pack, sel["tokens"], sel["num_ims"], ctx_prev, ctx_all, status=status, qkv=sel)

        if overlap:                                       # (side_stream already waits for this level's tokens / num_ims: fork_behind above)
            keepalive.append((sel["tokens"], sel["num_ims"], sel.get("qkv_img"), sel.get("_qkv_ws")))
            with torch.cuda.stream(side_stream), _Range(f"level {i}: aggregator (second stream)"):
                agg = aggregate()
        else:
            with _Range(f"level {i}: aggregator"):
                agg = aggregate()
        out = {"logits": agg["logits"], "ctx_slide": agg["ctx_slide"], "ctx_patch": sel["ctx_patch"], "importance": sel["importance"]}
        ctx_hist.append(out["ctx_slide"])
        rec = None
        if trace is not None:
            rec = {"num_ims": num_ims, "locs": locs, "parent_inds": parent_inds, "importance": out["importance"],
                   "logits": out["logits"], "ctx_slide": out["ctx_slide"]}
            trace.append(rec)
        if i == num_levels - 1:
            break
        keep = int(keep_patches[i])
        cap_keep = N if keep < 0 else min(N, keep)
        # After the top-K the chain forks: the kept parents' h-partials (gather + GEMM, the longer branch) stay on this stream,
        # the child expansion and the row gathers (tiny latency-bound kernels) run beside them on a third stream and are joined
        # before the next level's gate GEMMs.  (The fork travels as the top-K kernel's stop event: _lib.fork_behind.)
        forked = overlap and par_stream is not None and share_parent
        if fused_topk:
            # the importance finish of this level already selected (ops.FUSE_TOPK); the expansion stream waits on that kernel's stop event
            keep_idx, keep_count, kept_rows = sel["keep_idx"], sel["keep_count"], sel["kept_rows"]
            keepalive.append((keep_idx, keep_count, kept_rows))
        else:
            keep_idx = torch.empty((B, cap_keep), **i32)
            keep_count = torch.empty((B,), **i32)
            kept_rows = None
            with (_lib.fork_behind([par_stream], main_stream) if forked else contextlib.nullcontext()):
                if rows_in_place:
                    # ... with the addresses of the kept parents' h rows (row b, i -> ctx_patch[b, keep_idx[b, i], :D]) for the parent GEMM
                    kept_rows = torch.empty((B, cap_keep), **i64)
                    ops.timed("topk", lambda: _lib.call("paths_topk_rows", p(out["importance"]), N, p(num_ims), B, N, keep, p(keep_idx), cap_keep,
                                                        p(keep_count), p(out["ctx_patch"]), Dp, N, p(kept_rows), p(zero_row), st))
                else:
                    _lib.call("paths_topk", p(out["importance"]), N, p(num_ims), B, N, keep, p(keep_idx), cap_keep, p(keep_count), st)
        Nn = 4 * cap_keep
        hp = ops.parent_partials(lstm_pack, out["ctx_patch"], keep_idx, keep_count, kept_rows) if share_parent else None
        st2 = par_stream.cuda_stream if forked else st
        with (torch.cuda.stream(par_stream) if forked else contextlib.nullcontext()):
            def expand(cap):
                bufs = (torch.empty((B,), **i64), torch.empty((B, cap, 2), **i64), torch.empty((B, cap), **i64),
                        torch.empty((B, cap), **i32), torch.empty((B, cap), **i32), torch.empty((B, cap), **i32))
                _lib.call("paths_expand_children", p(keep_idx), cap_keep, p(keep_count), p(locs), N, mc.patch_size,
                          p(gx[i + 1]), p(gy[i + 1]), p(mask_ptrs[i + 1]), B, cap, p(bufs[0]), p(bufs[1]), p(bufs[2]),
                          p(bufs[3]), p(bufs[4]), p(status), None, p(bufs[5]) if share_parent else None, st2)
                return bufs

            num_next, locs_next, parent_next, src_row, src_cell, hp_row = ops.timed("expand", lambda: expand(Nn))
            if careful:
                empty = (num_next == 0).cpu()                      # per-level sync: slow path only
                if bool(empty.any()):
                    need = Nn
                    for b in torch.nonzero(empty).flatten().tolist():
                        tissue = int(batch.slides[b].masks[i + 1].sum().item())
                        X, Y = batch.slides[b].shape(i + 1)
                        need = max(need, tissue if tissue > 0 else X * Y)
                    if need > Nn:
                        Nn = need
                        num_next, locs_next, parent_next, src_row, src_cell, hp_row = expand(Nn)
                    _lib.call("paths_fallback_all_cells", p(gx[i + 1]), p(gy[i + 1]), p(mask_ptrs[i + 1]), mc.patch_size, B, Nn,
                              p(num_next), p(locs_next), p(parent_next), p(src_row), p(src_cell), p(status),
                              p(hp_row) if share_parent else None, st2)
            x_rows_next = None
            if share_parent:
                # children only need their parent's c row (h enters through the per-parent partials below)
                Hc = Dp - D
                state_next = torch.empty((B, Nn, Hc), **f32)
                if rows_in_place:
                    # ... and their feature rows are not copied either: the GEMMs of the next level read them in the resident grids
                    fts_next = None
                    x_rows_next = torch.empty((B, Nn), **i64)
                    ops.timed("gather", lambda: _lib.call(
                        "paths_gather_rows", p(grid_ptrs[i + 1]), p(src_cell), D, out["ctx_patch"].data_ptr() + 4 * D, N, Dp,
                        p(src_row), Hc, p(num_next), B, Nn, None, p(state_next), 0, p(x_rows_next), p(zero_row), st2))
                else:
                    fts_next = torch.empty((B, Nn, D), **f32)
                    _lib.call("paths_gather_rows", p(grid_ptrs[i + 1]), p(src_cell), D, out["ctx_patch"].data_ptr() + 4 * D, N, Dp,
                              p(src_row), Hc, p(num_next), B, Nn, p(fts_next), p(state_next), 0, None, None, st2)
                parent = {"hp": hp, "hp_row": hp_row, "c0": state_next}
                state_next = None
            else:
                fts_next = torch.empty((B, Nn, D), **f32)
                state_next = torch.empty((B, Nn, Dp), **f32)
                _lib.call("paths_gather_rows", p(grid_ptrs[i + 1]), p(src_cell), D, p(out["ctx_patch"]), N, Dp, p(src_row), Dp,
                          p(num_next), B, Nn, p(fts_next), p(state_next), 0, None, None, st2)
            if forked:
                fork_pending = True
                keepalive.append((num_next, locs_next, parent_next, src_row, src_cell, hp_row, parent, fts_next, x_rows_next))
        if rec is not None:
            rec["keep_idx"], rec["keep_count"] = keep_idx, keep_count
        fts, x_rows, locs, parent_inds, num_ims, state_prev, N = fts_next, x_rows_next, locs_next, parent_next, num_next, state_next, Nn
    if overlap:
        _lib.stream_wait(main_stream, side_stream)
        if par_stream is not None:
            _lib.stream_wait(main_stream, par_stream)
        keepalive.clear()
    out = dict(out)
    out["status"] = status
    return out


def recurse_train(model, slides, keep_patches: Sequence[int], num_levels: int, careful: bool = False) -> Dict[str, torch.Tensor]:
    """Differentiable recursion for training: same kernels as :func:`recurse`, but every level goes through
    paths_amd.autograd.LevelFn / GatherFn so that ``loss.backward()`` runs the hand-written backward kernels.
    Padded rows are zero-filled and computed (no tile skipping) so that every saved activation is finite.

    Like :func:`recurse` the default pass is optimistic and sync-free; the returned ``status`` word has bit 0 set when some
    slide's kept patches had no tissue children.  The caller (:func:`forward_backward`) then repeats the step with
    ``careful=True``: one host sync per level and the reference's fallback to all tissue cells of the next grid with zero
    parent state (data_utils/slide.py:336-352), handled on the device (paths_fallback_all_cells)."""
    batch = slides if isinstance(slides, DeviceSlideBatch) else DeviceSlideBatch(slides)
    with ops.range_guard(batch.feat_absmax):
        return _recurse_train_body(model, batch, keep_patches, num_levels, careful)


def _recurse_train_body(model, batch, keep_patches, num_levels, careful):
    from . import autograd as pag
    mc = model.procs[0].config
    ops.check_supported(mc, training=True)
    B, dev, D = len(batch), batch.device, batch.dim
    st = _lib.stream()
    p = _lib.ptr
    i32 = dict(device=dev, dtype=torch.int32)
    i64 = dict(device=dev, dtype=torch.int64)
    f32 = dict(device=dev, dtype=torch.float32)
    status = torch.zeros(1, **i32)
    N = batch.n0
    fts = torch.empty((B, N, D), **f32)
    locs = torch.empty((B, N, 2), **i64)
    parent = torch.empty((B, N), **i64)
    num_ims = torch.empty((B,), **i64)
    _lib.call("paths_level0_batch", p(batch.grid_ptrs[0]), p(batch.gx[0]), p(batch.gy[0]), B, D, mc.patch_size, N,
              p(fts), p(locs), p(parent), p(num_ims), 1, None, None, st)
    state_prev, ctx_prev, ctx_hist = None, None, []
    logits = None
    # Once-per-parent form (LSTM, optimistic pass): siblings share their parent's h, so the h half of the gate pre-activations is one
    # product over the kept parents and its gradients are products over a quarter of the rows (autograd.LevelParentFn /
    # GatherParentFn).  The careful re-run (rare: a slide whose kept patches have no tissue children) keeps the per-child form,
    # whose fallback rows carry no parent.  PATHS_TRAIN_PARENT=0 keeps the per-child form everywhere.
    parent_form = TRAIN_PARENT and model.use_lstm and not careful
    par = None
    for i in range(num_levels):
        if mc.slide_ctx_mode == "concat":          # the classifier reads every previous level's slide context (model/paths.py:134-137)
            ctx_prev = torch.stack(ctx_hist, dim=1) if ctx_hist else None
        if par is not None:
            logits, ctx_slide, state_out, importance = pag.LevelParentFn.apply(
                model.procs[i], model.lstm, fts, locs, num_ims, par["c0"], par["h_kept"], par["hp_row"], par["child_pos"], par["keep_count"],
                par["cap"], ctx_prev, *pag.lstm_params(model.lstm), *pag.level_params(model.procs[i]))
        else:
            logits, ctx_slide, state_out, importance = pag.level_apply(model.procs[i], model.lstm if model.use_lstm else None, fts, locs,
                                                                       num_ims, state_prev, ctx_prev)
        ctx_prev = ctx_slide
        ctx_hist.append(ctx_slide)
        if i == num_levels - 1:
            break
        keep = int(keep_patches[i])
        cap_keep = N if keep < 0 else min(N, keep)
        keep_idx = torch.empty((B, cap_keep), **i32)
        keep_count = torch.empty((B,), **i32)
        _lib.call("paths_topk", p(importance), N, p(num_ims), B, N, keep, p(keep_idx), cap_keep, p(keep_count), st)
        Nn = 4 * cap_keep
        num_next = torch.empty((B,), **i64)
        locs_next = torch.empty((B, Nn, 2), **i64)
        parent_next = torch.empty((B, Nn), **i64)
        src_row = torch.empty((B, Nn), **i32)
        src_cell = torch.empty((B, Nn), **i32)
        child_pos = torch.empty((B, 4 * cap_keep), **i32)
        hp_row = torch.empty((B, Nn), **i32) if parent_form else None

        def expand():
            _lib.call("paths_expand_children", p(keep_idx), cap_keep, p(keep_count), p(locs), N, mc.patch_size,
                      p(batch.gx[i + 1]), p(batch.gy[i + 1]), p(batch.mask_ptrs[i + 1]), B, Nn, p(num_next), p(locs_next),
                      p(parent_next), p(src_row), p(src_cell), p(status), p(child_pos), p(hp_row), st)

        expand()
        if careful:
            empty = (num_next == 0).cpu()                      # per-level sync: slow path only
            if bool(empty.any()):
                need = Nn
                for b in torch.nonzero(empty).flatten().tolist():
                    tissue = int(batch.slides[b].masks[i + 1].sum().item())
                    X, Y = batch.slides[b].shape(i + 1)
                    need = max(need, tissue if tissue > 0 else X * Y)
                if need > Nn:
                    Nn = need
                    locs_next = torch.empty((B, Nn, 2), **i64)
                    parent_next = torch.empty((B, Nn), **i64)
                    src_row = torch.empty((B, Nn), **i32)
                    src_cell = torch.empty((B, Nn), **i32)
                    expand()
                # fallback rows: src_row = -1 (zero parent state, no gradient to any parent: child_pos of that slide is all -1)
                _lib.call("paths_fallback_all_cells", p(batch.gx[i + 1]), p(batch.gy[i + 1]), p(batch.mask_ptrs[i + 1]), mc.patch_size,
                          B, Nn, p(num_next), p(locs_next), p(parent_next), p(src_row), p(src_cell), p(status), None, st)
        if parent_form:
            fts, c0, h_kept = pag.GatherParentFn.apply(state_out, batch.grid_ptrs[i + 1], src_cell, src_row, num_next, keep_idx, keep_count,
                                                       child_pos, D, Nn)
            par = {"c0": c0, "h_kept": h_kept, "hp_row": hp_row, "child_pos": child_pos, "keep_count": keep_count, "cap": cap_keep}
            state_prev = None
        else:
            fts, state_prev = pag.GatherFn.apply(state_out, batch.grid_ptrs[i + 1], src_cell, src_row, num_next, keep_idx,
                                                 keep_count, child_pos, D, Nn)
        locs, parent, num_ims, N = locs_next, parent_next, num_next, Nn
    return {"logits": logits, "status": status}


def _label_to(x, dev):
    """A label vector on the device without a blocking copy: a pageable-memory upload would hold the host until everything already
    enqueued on the stream (the whole forward) has run, and the backward could only be enqueued after that."""
    t = torch.as_tensor(x)
    if t.device == dev:
        return t
    if dev.type == "cuda" and not t.is_cuda:
        return t.pin_memory().to(dev, non_blocking=True)
    return t.to(dev)


def loss_from_logits(logits, batch, task: str, global_batch: Optional[int] = None):
    """Loss of reference utils.py:263-279 on the last level's logits.  With ``global_batch`` the mean is taken over the
    GLOBAL batch (sum of local terms / global_batch) so that data-parallel shards add up exactly."""
    dev = logits.device
    n = logits.shape[0]
    scale = 1.0 if global_batch is None else n / float(global_batch)
    if task == "survival":
        hazards = torch.sigmoid(logits)
        loss = nll_loss(hazards, _label_to(batch["survival_bin"], dev), _label_to(batch["censored"], dev))
        return hazards, loss * scale
    elif task == "subtype_classification":
        return logits, F.cross_entropy(logits, _label_to(batch["subtype"], dev)) * scale
    raise ValueError(task)


_STATUS_HOST: Dict[int, torch.Tensor] = {}


def forward_backward(model, batch, num_levels, keep_patches, task: str = "survival", global_batch: Optional[int] = None):
    """Forward recursion + loss + backward of one (local) batch, with the recursion's status word checked BEFORE any gradient
    leaves the rank (all-reduce / optimizer): the status is copied to pinned memory right after the forward and read once the
    backward has been enqueued (by then the forward has long finished, so the wait is free).  Bit 0 (a slide without tissue
    children) repeats forward + backward on the careful path (reference fallback, data_utils/slide.py:336-352); bit 2
    (capacity) raises.  Gradients must be clear (set to None) on entry.  Returns (outputs, loss)."""
    from . import autograd as pag
    out = recurse_train(model, batch["slide"], keep_patches, num_levels)
    dev = out["status"].device
    host = _STATUS_HOST.get(dev.index)
    if host is None:
        host = _STATUS_HOST[dev.index] = torch.zeros((1,), dtype=torch.int32).pin_memory()
    host.copy_(out["status"], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    outputs, loss = loss_from_logits(out["logits"], batch, task, global_batch)
    loss.backward()
    ev.synchronize()
    if check_status_word(int(host[0])):
        model.zero_grad(set_to_none=True)
        out = recurse_train(model, batch["slide"], keep_patches, num_levels, careful=True)
        outputs, loss = loss_from_logits(out["logits"], batch, task, global_batch)
        loss.backward()
        check_status_word(out["status"], fallback_done=True)
    pag.fill_dead_grads(model)
    return outputs, loss


def train_step(model, optimizer, batch, num_levels, keep_patches, task: str = "survival", global_batch: Optional[int] = None,
               allreduce=None):
    """One optimisation step with the reference's semantics (train.py:59-68): forward recursion, mean loss,
    backward, [gradient all-reduce], optimizer step.  Dead parameters get the reference's zero gradients.
    ``batch`` None = this rank holds no slide of a short global batch: it contributes zeros for exactly the gradient set of
    an active rank.  Returns the (local share of the) loss as a tensor (None for an idle rank)."""
    from . import autograd as pag
    if ZERO_GRAD_ALL:
        optimizer.zero_grad(set_to_none=True)
    else:
        pag.clear_grads(model, optimizer)          # (the same for the live parameters; the dead ones keep their shared zero views)
    loss = None
    if batch is None:
        pag.zero_live_grads(model, num_levels)
    else:
        _, loss = forward_backward(model, batch, num_levels, keep_patches, task, global_batch)
        loss = loss.detach()
    if allreduce is not None:
        allreduce(model, num_levels=num_levels)
    optimizer.step()
    return loss


def inference_end2end(num_levels, keep_patches, model, base_power, batch, task: str):
    """reference utils.py:228-279.  ``batch["slide"]`` is a list of :class:`DeviceSlide`; labels as in the reference
    (``survival_bin`` / ``censored`` or ``subtype``).  Returns (hazards or logits, loss)."""
    slides = batch["slide"]
    dev = slides.device if isinstance(slides, DeviceSlideBatch) else slides[0].grids[0].device
    out = recurse(model, slides, keep_patches, num_levels)
    logits = out["logits"]
    if task == "survival":
        labels = torch.as_tensor(batch["survival_bin"]).to(dev)
        censors = torch.as_tensor(batch["censored"]).to(dev)
        hazards = torch.sigmoid(logits)
        return hazards, nll_loss(hazards, labels, censors)
    elif task == "subtype_classification":
        subtypes = torch.as_tensor(batch["subtype"]).to(dev)
        return logits, F.cross_entropy(logits, subtypes)
    raise ValueError(task)
