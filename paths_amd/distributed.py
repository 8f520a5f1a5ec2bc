"""Slide-sharded data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" for the CPU tests).

The forward path shards naturally — a slide's recursion touches only its own rows (reference utils.py:252-258) —
so there is NO data-path collective: ranks own disjoint slices of the slide batch, run the recursion
independently, and only (a) barriers around a timed region, (b) a MAX-reduce of elapsed time and (c) an
all-gather of the tiny per-slide outputs ([B, nbins] hazards) cross the fabric.  Training adds exactly one collective per
step, AFTER the backward has finished: :func:`allreduce_gradients`, one flat-bucket all-reduce(sum) of the live gradients
(not overlapped with the backward - the shared LSTM's gradient, more than half of the message, is only
complete when the last backward call returns).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str, device: Optional[torch.device] = None, force: bool = False) -> Tuple[int, int]:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).  A single rank needs no
    group; ``force`` (or PATHS_FORCE_DIST=1) creates one anyway, so that the collective code paths (RCCL with backend "nccl")
    can be exercised and timed on a one-GPU box."""
    rank, world, _ = env_rank_world()
    force = force or os.environ.get("PATHS_FORCE_DIST", "0") != "0"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n_total: int, rank: int, world: int) -> range:
    """Contiguous balanced slice of ``range(n_total)`` owned by ``rank`` (first ``n_total % world`` ranks get one more)."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float) -> List[float]:
    """Every rank's ``value`` in rank order, on every rank (per-rank step times: the load imbalance of variable child counts)."""
    if not dist.is_initialized():
        return [float(value)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, float(value))
    return [float(v) for v in out]


def gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-slide rows (e.g. hazards [b_local, nbins]) into global slide order on every rank."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = len(shard_range(n_total, 0, world))
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: len(shard_range(n_total, r, world))] for r in range(world)], dim=0)


LAST_ALLREDUCE_EVENTS = None     # (start, end) CUDA events around the most recent gradient all-reduce (bench.py --mode train)
TIME_ALLREDUCE = False


def allreduce_gradients(model, average: bool = False, num_levels: Optional[int] = None):
    """ONE flat-bucket all-reduce(sum) of the live gradients (RCCL over xGMI with backend "nccl").
    The loss of each rank is already scaled by local_batch / global_batch (paths_amd.utils.loss_from_logits), so the sum
    of the shards' gradients IS the gradient of the global-batch mean loss.  The bucket walks a FIXED parameter list
    (paths_amd.autograd.live_grad_params: identical on every rank, whatever slides a rank holds), so the message has the
    same element count everywhere; a missing gradient in that list counts as zeros.  The dead parameters (nn.Transformer
    encoder, cross-attention matrices: zero gradients on every rank) are not sent: 7.2 M fp32 = 29 MB per step instead of
    39.5 MB.  The classifiers of the non-final levels have grad None on every rank and stay None."""
    if not dist.is_initialized():
        return
    from . import autograd as pag
    params = pag.live_grad_params(model, num_levels)
    flat, slices = _grad_bucket(model, params)
    # One message per step, in PERSISTENT storage (round 4 allocated a 29 MB torch.cat result per step).  Gradients that are not
    # already the bucket's own slices are copied in by one multi-tensor copy; a missing gradient counts as zeros.
    todo_dst, todo_src = [], []
    for p, sl in zip(params, slices):
        g = p.grad
        if g is None:
            sl.zero_()
        elif g.data_ptr() != sl.data_ptr():
            todo_dst.append(sl)
            todo_src.append(g.reshape(-1))
    if todo_dst:
        torch._foreach_copy_(todo_dst, todo_src)
    global LAST_ALLREDUCE_EVENTS
    ev = None
    if TIME_ALLREDUCE and flat.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dist.get_backend() == "gloo" and flat.is_cuda:         # CPU-collective rehearsal path (tests / 1-GPU boxes)
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if ev is not None:
        ev[1].record()
        LAST_ALLREDUCE_EVENTS = ev
    if average:
        flat /= dist.get_world_size()
    # the reduced bucket BECOMES the gradients: every .grad is re-pointed at its slice of the flat buffer (no copy back: round 3
    # issued one copy_ launch per parameter here, ~150 launches on a host-bound step).  NOTE: these .grad tensors alias ONE
    # storage (16-byte aligned slices, so the one-launch AdamW keeps its float4 path); ``zero_grad(set_to_none=True)`` - what
    # paths_amd.train and the reference's loop use - drops the views, ``set_to_none=False`` zeroes the bucket slice by slice.
    for p, sl in zip(params, slices):
        p.grad = sl.view_as(p)


_BUCKET_ALIGN = 4        # floats: every slice starts on a 16-byte boundary


def _grad_bucket(model, params):
    """(flat, slices) - the model's persistent gradient bucket for this parameter list: each parameter's slice starts at an offset
    rounded up to 4 floats (the padding words are zero and stay zero through the all-reduce)."""
    key = (tuple(id(p) for p in params), params[0].device if params else None)
    ent = getattr(model, "_paths_grad_bucket", None)
    if ent is None or ent[0] != key:
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + _BUCKET_ALIGN - 1) // _BUCKET_ALIGN * _BUCKET_ALIGN
        flat = torch.zeros((max(off, _BUCKET_ALIGN),), dtype=torch.float32, device=key[1])
        slices = [flat[o:o + p.numel()] for o, p in zip(offs, params)]
        ent = (key, flat, slices)
        object.__setattr__(model, "_paths_grad_bucket", ent)
    return ent[1], ent[2]
