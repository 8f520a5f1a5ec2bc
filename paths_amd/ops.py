"""Host-side launch sequence of one magnification level on the HIP kernels (no math in Python).

``level_forward`` is what both the drop-in ``PATHSProcessor.process`` (paths_amd/model/paths.py) and the
device-resident recursion (paths_amd/utils.py) call.  It only allocates device buffers through PyTorch's
caching allocator and issues C-ABI calls on the current stream; nothing synchronises.

Per level (B slides, N padded rows, D features, T = N+1 tokens) the launch sequence is

    paths_lstm_cell          3 launches   gates GEMM (c part, o part) + mem_to_out GEMM, fused epilogues
    paths_importance_proj    1 launch     importance MLP + sigmoid + mask + proj_in + PE + special token
    paths_token_layer_f32    1 launch     in_proj of layer 0
    per layer l < L-1:  paths_attention_f32 + paths_token_layer_f32 (post-attention chain + in_proj of l+1)
    paths_token0_tail        1 launch     last layer at token 0 only + decoder.norm + slide ctx residual + classifier
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib

LOG2E = 1.4426950408889634
QKV_IMAGES = os.environ.get("PATHS_QKV_IMAGES", "1") != "0"
TAIL_WS = os.environ.get("PATHS_TAIL_WS", "1") != "0"          # token-0 tail without K / V projections, one launch (csrc/token0_ws.hip)
TLAYER_WS = os.environ.get("PATHS_TLAYER_WS", "1") != "0"      # weight-stationary token-layer kernel (csrc/tlayer_ws.hip)
SPLITK_IMPORTANCE = os.environ.get("PATHS_SPLITK_IMPORTANCE", "1") != "0"
# The first decoder layer's in_proj inside the finish of the importance / projection GEMM (paths_importance_qkv_x6, round 5): 1 = one
# fused finish on the selection stream (importance + tokens + q | k | v operand images), 2 (default) = importance-only finish on the selection
# stream, tokens + images on the aggregator stream, 0 = the round-4 form (finish, then paths_token_layer_ws as the aggregator's first launch)
FUSE_QKV = int(os.environ.get("PATHS_FUSE_QKV", "2"))
# ... and (FUSE_QKV = 2, device recursion) the top-K of the level inside that importance finish: one launch instead of two tiny ones on the
# recursion's critical path (paths_importance_qkv_x6 phase 8).  Built, bit-identical to the separate launch - and SLOWER: 23.5 us against
# 7.0 + 9.7 (tools/fin_time.py): a device-side arrival barrier (write-through stores, drain, release add, polling, acquire, L2-bypassing
# re-reads of 2,048 scores per workgroup) costs more than the kernel boundary it replaces.  Off by default; kept for A/B runs.
FUSE_TOPK = os.environ.get("PATHS_FUSE_TOPK", "0") != "0"
KERNEL_TIMER = None   # optional hook: fn(name, launch_callable, meta) — set by bench.py only
TIMER_ALL = False     # with KERNEL_TIMER: False = bracket only the dominant kernel and the aggregator span (the timed region of
                      # bench.py), True = every kernel group (bench.py's serialised breakdown pass)


def timed(name: str, fn, meta=None, detail: bool = True):
    """Run ``fn`` (a launch sequence on the current stream); bench.py may bracket it with events.  ``detail`` groups are only
    bracketed in the breakdown pass."""
    if KERNEL_TIMER is None or (detail and not TIMER_ALL):
        return fn()
    return KERNEL_TIMER(name, fn, meta or {})
# Which matrix pipe the big products (selection-chain GEMMs, attention) use.  All are HIP kernels of libpaths_hip.so with fp32
# inputs, outputs and accumulation:
#   "h3"  : operands split into TWO fp16 planes, 3 MFMAs per product block (csrc/gemm_x6.hip, NP = 2)            - default
#           weights are pre-scaled per tensor by a power of two, activations by A_SCALE (|activation| must stay < 65504 / A_SCALE)
#   "x6"  : operands split into THREE bf16 planes, 6 MFMAs per product block (same kernels, NP = 3; no range limits)
#   "f32" : f32-input MFMA (csrc/gemm_f32.hip, csrc/attn_f32.hip)
GEMM_MODE = os.environ.get("PATHS_GEMM_MODE", "h3")
A_SCALE = float(os.environ.get("PATHS_H3_A_SCALE", "16"))      # a power of two
AGG_FP8 = os.environ.get("PATHS_AGG_FP8", "0") != "0"       # the whole aggregator's big products (in_proj, attention, out_proj, FFN) with e4m3 operands: the BASELINE configs[4] stress variant (opt-in, NOT a parity path: csrc/gemm_fp8.hip)
ATTN_FP8 = os.environ.get("PATHS_ATTN_FP8", "0") != "0"     # inference attention with e4m3 operands (opt-in, see csrc/attn_fp8.hip)
# Operand split of the training step's GRADIENT GEMMs (dX = dY W, dW = dY^T X), bf16 planes (fp32's exponent range: gradients span too
# many binades for a fixed-scale fp16 split, and re-imaging weights every step must not cost a host sync on max|w|):
#   4 (default) = TWO planes (hi + mid, 16 significant bits per operand, 3 MFMAs per product block, relative error ~2e-5 per product);
#   3           = three planes (exact fp32 products, 6 MFMAs).
TRAIN_PLANES = int(os.environ.get("PATHS_TRAIN_PLANES", "4"))
assert TRAIN_PLANES in (3, 4), "PATHS_TRAIN_PLANES must be 3 (exact) or 4 (two bf16 planes)"
# forward GEMMs / attention of the TRAINING step: 2 = the inference split (fp16 planes) with lagged weight scales, 3 = exact bf16
TRAIN_FWD_PLANES = int(os.environ.get("PATHS_TRAIN_FWD_PLANES", "2"))


H3_STATE_MARGIN = 8.0   # |h1| <= 1 and |c1| <= depth + 1 by construction (sigmoid * tanh, interface.py:52-56): x + h1, c stay inside


def h3_in_range(feat_absmax: float) -> bool:
    """The fp16-split operands of the default mode hold |activation| * A_SCALE: features up to (65504 / A_SCALE) - margin."""
    return math.isfinite(feat_absmax) and (feat_absmax + H3_STATE_MARGIN) * A_SCALE < 65504.0


RANGE_FALLBACKS = [0]    # how often a batch left the fp16 range and ran on the bf16 split (diagnostic / tests)


class range_guard:
    """``with range_guard(max|feature|):`` — the range contract of the default fp16-split mode, enforced where data enters the
    path (every resident grid's max|x| comes with its tissue-mask pass, drop-in batches are reduced on entry): inside the block
    an out-of-range batch runs on the exact three-plane bf16 kernels (PATHS_GEMM_MODE semantics "x6": same entry points, no
    range limits, ~1.6x slower) instead of silently producing inf/NaN planes; non-finite features raise."""

    def __init__(self, feat_absmax: float):
        self.amax = float(feat_absmax)
        self.saved = None

    def __enter__(self):
        global GEMM_MODE, TRAIN_FWD_PLANES
        if not math.isfinite(self.amax):
            raise _lib.PathsHipError("input features contain inf or NaN")
        if not h3_in_range(self.amax) and (GEMM_MODE == "h3" or TRAIN_FWD_PLANES == 2):
            self.saved = (GEMM_MODE, TRAIN_FWD_PLANES)
            RANGE_FALLBACKS[0] += 1
            GEMM_MODE, TRAIN_FWD_PLANES = ("x6" if GEMM_MODE == "h3" else GEMM_MODE), 3
        return self

    def __exit__(self, *exc):
        global GEMM_MODE, TRAIN_FWD_PLANES
        if self.saved is not None:
            GEMM_MODE, TRAIN_FWD_PLANES = self.saved
        return False


def split_planes() -> int:
    if GEMM_MODE not in ("h3", "x6", "f32"):
        raise ValueError(f"PATHS_GEMM_MODE must be 'h3', 'x6' or 'f32' (got {GEMM_MODE!r})")
    return 2 if GEMM_MODE == "h3" else 3


def a_scale() -> float:
    return A_SCALE if GEMM_MODE == "h3" else 1.0


def x6_pack(w: torch.Tensor, n_pad: Optional[int] = None, planes: Optional[int] = None, w_scale: Optional[float] = None):
    """fp32 [N, K] device weight -> (split tiled image as a byte tensor, w_scale)  (paths_x6_pack_weights).
    planes 3: exact bf16 hi|mid|lo, w_scale 1.  planes 2: fp16 hi|lo of w * w_scale, w_scale = the power of two that puts
    max|w| into [8192, 16384) (one host sync per weight version)."""
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1
    planes = split_planes() if planes is None else planes
    N, K = w.shape
    n_pad = N if n_pad is None else n_pad
    if planes != 2:
        w_scale = 1.0
    elif w_scale is None:
        w_scale = _pow2_scale(w)
    out = torch.empty((n_pad * K * 2 * (2 if planes == 4 else planes),), device=w.device, dtype=torch.uint8)
    _lib.call("paths_x6_pack_weights", _lib.ptr(w), w.stride(0), _lib.ptr(out), N, n_pad, K, planes, w_scale, _lib.stream())
    return out, w_scale


def _pow2_scale(w: torch.Tensor) -> float:
    """The power of two that puts max|w| into [8192, 16384) (fp16 operand scaling; one host sync)."""
    amax = float(w.abs().max())
    return 2.0 ** max(-14, min(24, math.floor(math.log2(16384.0 / amax)))) if amax > 0 and math.isfinite(amax) else 1.0


_LAGGED_EPOCH = [0]      # bumped by reset_lagged_scales(): every per-module store older than this is discarded on next use


def _lagged_store(owner) -> Dict[object, list]:
    """The lagged-scale entries of one module live ON the module (not in a table keyed by id(), which Python may hand to
    another module after this one is collected)."""
    st = owner.__dict__.get("_paths_lagged")
    if st is None or st.get("_epoch") != _LAGGED_EPOCH[0]:
        st = {"_epoch": _LAGGED_EPOCH[0]}
        object.__setattr__(owner, "_paths_lagged", st)
    return st


def _pow2_scale_lagged(w: torch.Tensor, owner, key) -> float:
    """:func:`_pow2_scale` without a host sync after the first call per ``key``: training re-images its weights every step, so
    the scale in use is the one computed from an EARLIER version of the same weight (its max|w| is reduced on the device, copied
    to pinned memory asynchronously and picked up once the copy has landed).  The scale leaves a factor 4 of fp16 headroom above
    max|w|, which no optimizer step at the reference's learning rates can cross between two refreshes."""
    store = _lagged_store(owner)
    ent = store.get(key)
    if ent is None:
        ent = store[key] = [_pow2_scale(w), None, torch.empty((1,), dtype=torch.float32).pin_memory()]
    elif ent[1] is not None and ent[1].query():
        amax = float(ent[2][0])
        if not math.isfinite(amax) or amax * ent[0] >= 65504.0:
            # the scale used since the last refresh was too large for this weight (or the weight is not finite): the fp16 planes
            # built with it overflowed.  Loud, if late; weights replaced wholesale must call reset_lagged_scales() (load_state does)
            store.pop(key, None)
            raise _lib.PathsHipError(f"weight {key!r}: max|w| = {amax} left the fp16 range of its lagged scale {ent[0]}; "
                                     "call paths_amd.ops.reset_lagged_scales() after replacing weights")
        if amax > 0:
            ent[0] = 2.0 ** max(-14, min(24, math.floor(math.log2(16384.0 / amax))))
        ent[1] = None
    if ent[1] is None:
        ent[2].copy_(w.detach().abs().max().reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ent[1] = ev
    return ent[0]


def reset_lagged_scales():
    """Forget the lagged weight scales (next use recomputes them with a host sync): call after loading a checkpoint or any other
    wholesale replacement of the weights of a model that is being trained."""
    _LAGGED_EPOCH[0] += 1


def tlayer_h3_images(layer: Dict[str, object], part: int):
    """(image, scales) of a decoder layer's weights for paths_token_layer_h3: part 0 = (wo, w1, w2), part 1 = wqkv.
    Built on first use and cached in the layer's pack dict (rebuilt when weights change)."""
    key = f"h3_image_{part}"
    if key not in layer:
        ws = [layer["wo"], layer["w1"], layer["w2"]] if part == 0 else [layer["wqkv"]]
        scales = tuple(_pow2_scale(w) for w in ws)
        nbytes = int(_lib.load().paths_tlayer_h3_image_bytes(part))
        img = torch.empty((nbytes,), device=ws[0].device, dtype=torch.uint8)
        pw = [_lib.ptr(w) for w in ws] + [None] * (3 - len(ws))
        sc = list(scales) + [1.0] * (3 - len(scales))
        _lib.call("paths_tlayer_pack_h3", part, pw[0], pw[1], pw[2], sc[0], sc[1], sc[2], _lib.ptr(img), _lib.stream())
        layer[key] = (img, scales)
    return layer[key]


def tlayer_ws_images(layer: Dict[str, object], part: int):
    """(image, scales) of a decoder layer's weights for paths_token_layer_ws: part 0 = (wo, w1, w2), part 1 = wqkv; cached in the
    layer's pack dict like :func:`tlayer_h3_images`."""
    key = f"ws_image_{part}"
    if key not in layer:
        ws = [layer["wo"], layer["w1"], layer["w2"]] if part == 0 else [layer["wqkv"]]
        scales = tuple(_pow2_scale(w) for w in ws)
        d = layer["wo"].shape[0]
        img = torch.empty((int(_lib.load().paths_tlayer_ws_image_bytes(part, d)),), device=ws[0].device, dtype=torch.uint8)
        pw = [_lib.ptr(w) for w in ws] + [None] * (3 - len(ws))
        sc = list(scales) + [1.0] * (3 - len(scales))
        _lib.call("paths_tlayer_pack_ws", part, pw[0], pw[1], pw[2], sc[0], sc[1], sc[2], _lib.ptr(img), d, _lib.stream())
        layer[key] = (img, scales)
    return layer[key]


def token0_ws_image(layer: Dict[str, object], qscale: float) -> torch.Tensor:
    """Weight image of paths_token0_tail_ws for the LAST decoder layer (cached in the layer's pack dict)."""
    if "t0_image" not in layer:
        d = int(layer["wo"].shape[0])          # trans_dim: 128 or 192 (csrc/token0_ws.hip)
        img = torch.empty((int(_lib.load().paths_token0_ws_image_bytes_d(d)),), device=layer["wo"].device, dtype=torch.uint8)
        _lib.call("paths_token0_pack_ws_d", _lib.ptr(layer["wqkv"]), _lib.ptr(layer["bqkv"]), _lib.ptr(layer["wo"]), _lib.ptr(layer["bo"]),
                  _lib.ptr(layer["w1"]), _lib.ptr(layer["w2"]), qscale, d, _lib.ptr(img), _lib.stream())
        layer["t0_image"] = img
    return layer["t0_image"]


_T0_COUNTERS: Dict[tuple, torch.Tensor] = {}
_T0_RETIRED: List[torch.Tensor] = []


_TOPK_COUNTERS: Dict[tuple, torch.Tensor] = {}


def topk_counters(dev, B: int) -> torch.Tensor:
    """Arrival / departure counters of the fused importance + top-K finish (paths_importance_qkv_x6 phase 8): 2 int32 words per slide
    that are zero between launches; one buffer per (device, stream), never handed back (recorded launch tapes hold its address)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _lib.stream())
    t = _TOPK_COUNTERS.get(key)
    if t is None or t.numel() < 2 * B:
        if t is not None:
            _T0_RETIRED.append(t)
        t = _TOPK_COUNTERS[key] = torch.zeros((max(512, 2 * B),), device=dev, dtype=torch.int32)
    return t


def token0_counters(dev, B: int) -> torch.Tensor:
    """Arrival tickets / flags of paths_token0_tail_ws: 3 int32 words per slide that are zero between launches (the last arrivers
    reset them).  One buffer per (device, stream): tails of different streams may be in flight together."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _lib.stream())
    t = _T0_COUNTERS.get(key)
    if t is None or t.numel() < 3 * B:
        if t is not None:
            _T0_RETIRED.append(t)        # recorded launch tapes / captured graphs hold its address: never hand the block back
        t = _T0_COUNTERS[key] = torch.zeros((max(768, 3 * B),), device=dev, dtype=torch.int32)
    return t


def _x6_of(pack: Dict[str, object], key: str, planes: Optional[int] = None, lagged: bool = False):
    """(image, w_scale) of pack[key] for the given (default: current) split, built on first use and cached beside it (the pack
    dict is rebuilt when weights change).  ``lagged`` (training, two-plane split): see :func:`_pow2_scale_lagged`."""
    planes = split_planes() if planes is None else planes
    k6 = f"{key}_split{planes}"
    if k6 not in pack:
        ws = _pow2_scale_lagged(pack[key], pack["_owner"], key) if (lagged and planes == 2) else None
        pack[k6] = x6_pack(pack[key], planes=planes, w_scale=ws)
    return pack[k6]


def pe_table(lvl_pack: Dict[str, object], pe_mode: int, d: int, rows: int) -> torch.Tensor:
    """[cap >= rows, d/2 or d] table of the positional-encoding sin/cos values (paths_pe_table), cached beside the weights."""
    div = lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]
    key = (pe_mode, d, div.device)
    tab = _PE_TABLES.get(key)
    if tab is None or tab.shape[0] < rows:
        cap = 1 << max(6, (rows - 1).bit_length())
        tab = torch.empty((cap, d // 2 if pe_mode == 2 else d), device=div.device, dtype=torch.float32)
        _lib.call("paths_pe_table", _lib.ptr(div), pe_mode, d, cap, _lib.ptr(tab), _lib.stream())
        if key in _PE_TABLES:
            _PE_RETIRED.append(_PE_TABLES[key])      # a recorded launch tape may still hold the smaller table's address
        _PE_TABLES[key] = tab
    return tab


# The positional-encoding constants depend on (d, device) only, not on the weights: kept across re-packs (training re-packs every
# step; two pageable host-to-device copies per level per step were 2.7 ms of a 25 ms step).
_PE_TABLES: Dict[tuple, torch.Tensor] = {}
_PE_RETIRED: List[torch.Tensor] = []
_PE_DIVS: Dict[tuple, tuple] = {}


def _pe_divs(d: int, dev) -> tuple:
    """(div_1d, div_2d): same expressions as reference utils.py:18 / :56, evaluated on the CPU like the CPU reference."""
    key = (d, dev)
    if key not in _PE_DIVS:
        k = 10000.0
        _PE_DIVS[key] = (torch.exp(torch.arange(0, d, 2) * (-math.log(k) / d)).float().to(dev),
                         torch.exp(torch.arange(0, d // 2, 2) * (-math.log(k) / d)).float().to(dev))
    return _PE_DIVS[key]


def use_x6(D: int, Hc: int = 64) -> bool:
    """Split-operand GEMMs (either split) are used for this shape."""
    split_planes()
    return GEMM_MODE != "f32" and D % 256 == 0 and Hc % 64 == 0


# ---------------------------------------------------------------------------------------------
# weight packing (cached on the owning module, invalidated by parameter version counters)
# ---------------------------------------------------------------------------------------------
def _versions(params):
    return tuple((p.data_ptr(), p._version) for p in params)


def pack_lstm(lstm) -> Dict[str, torch.Tensor]:
    """Gate rows regrouped so that one 96-column wave tile holds forget|remember|map of 32 memory units."""
    srcs = [lstm.forget_gate[0], lstm.remember_gate[0], lstm.remember_map[0], lstm.out_select_gate[0], lstm.mem_to_out[0]]
    params = getattr(lstm, "_paths_param_list", None)
    if params is None:
        params = [t for m in srcs for t in (m.weight, m.bias)]
        object.__setattr__(lstm, "_paths_param_list", params)
    key = _versions(params)
    cache = getattr(lstm, "_paths_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        f, r, m, o, mo = srcs
        Hc = f.weight.shape[0]
        assert Hc % 64 == 0, "hierarchical_ctx_mlp_hidden_dim must be a multiple of 64 for this build"
        wc = torch.stack([f.weight.view(Hc // 32, 32, -1), r.weight.view(Hc // 32, 32, -1), m.weight.view(Hc // 32, 32, -1)], dim=1)
        bc = torch.stack([f.bias.view(Hc // 32, 32), r.bias.view(Hc // 32, 32), m.bias.view(Hc // 32, 32)], dim=1)
        packed = {
            "w_gates": torch.cat([wc.reshape(3 * Hc, -1), o.weight], dim=0).float().contiguous(),
            "b_gates": torch.cat([bc.reshape(3 * Hc), o.bias], dim=0).float().contiguous(),
            "w_mem": mo.weight.detach().float().contiguous(),
            "b_mem": mo.bias.detach().float().contiguous(),
            "Hc": Hc,
            "_owner": lstm,
        }
    lstm._paths_pack = (key, packed)
    return packed


def pack_level(proc) -> Dict[str, object]:
    """Per-level tensors in the layout the kernels read.  Dead encoder / cross-attention matrices are
    never touched (only ``multihead_attn.out_proj.bias`` is live)."""
    agg = proc.global_agg
    params = getattr(proc, "_paths_param_list", None)        # (walking the module tree here was 1.5 ms of host time per training step)
    if params is None:
        params = list(proc.importance_mlp.parameters()) + [agg.proj_in.weight, agg.proj_in.bias, agg.special_token]
        params += list(agg.transformer.decoder.parameters()) + list(proc.classification_layer.parameters())
        if hasattr(proc, "hctx_mlp"):
            params += list(proc.hctx_mlp.parameters())
        object.__setattr__(proc, "_paths_param_list", params)
    key = _versions(params)
    cache = getattr(proc, "_paths_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        c = lambda t: t.detach().float().contiguous()
        d = agg.dim
        layers = _pack_decoder_layers(agg)
        dev = agg.proj_in.weight.device
        div_1d, div_2d = _pe_divs(d, dev)
        packed = {
            "w_ip": torch.cat([proc.importance_mlp[0].weight, agg.proj_in.weight], dim=0).float().contiguous(),
            # forward kernels: rows interleaved in blocks of 64 so that each column half of a workgroup owns 64 hidden units
            # and 64 token channels (csrc/gemm_epi.h EpiImpProj)
            "w_ip_fwd": torch.cat([proc.importance_mlp[0].weight[:64], agg.proj_in.weight[:64],
                                   proc.importance_mlp[0].weight[64:], agg.proj_in.weight[64:]], dim=0).float().contiguous(),
            "b1": c(proc.importance_mlp[0].bias), "w2": c(proc.importance_mlp[2].weight.view(-1)),
            "b2": c(proc.importance_mlp[2].bias.view(-1)),        # read on the device (a .item() here was a host sync per re-pack)
            "bp": c(agg.proj_in.bias), "special": c(agg.special_token),
            "div_1d": div_1d, "div_2d": div_2d,
            "layers": layers,
            "lnfg": c(agg.transformer.decoder.norm.weight), "lnfb": c(agg.transformer.decoder.norm.bias),
            "lnf_eps": float(agg.transformer.decoder.norm.eps),
            "wcls": c(proc.classification_layer.weight), "bcls": c(proc.classification_layer.bias),
            "_owner": proc,
        }
        if hasattr(proc, "hctx_mlp"):      # lstm=false: RNN hierarchical context (reference model/paths.py:49-54)
            packed.update({"wh1": c(proc.hctx_mlp[0].weight), "bh1": c(proc.hctx_mlp[0].bias),
                           "wh2": c(proc.hctx_mlp[2].weight), "bh2": c(proc.hctx_mlp[2].bias)})
    proc._paths_pack = (key, packed)
    return packed


def _pack_decoder_layers(agg) -> List[Dict[str, object]]:
    """Per-layer tensors of the aggregator's decoder stack in the kernels' naming (only ``multihead_attn.out_proj.bias`` of the
    cross-attention is live)."""
    c = lambda t: t.detach().float().contiguous()
    layers = []
    d, H = agg.dim, agg.nhead
    hd = d // H
    hdp = padded_head_dim(hd)
    pad = None
    if hdp != hd:
        # zero-padded heads (see padded_head_dim): "wqkv" [3 H hdp, d], "bqkv" [3 H hdp], "wo" [d, H hdp]; "_unpad" = where the true rows /
        # columns sit, for the gradients' way back (paths_amd/autograd.py:_level_grads)
        dev = agg.proj_in.weight.device
        col = (torch.arange(d, device=dev) // hd) * hdp + torch.arange(d, device=dev) % hd          # true inner index -> padded inner index
        row = torch.cat([col + k * H * hdp for k in range(3)])
        pad = (row, col, H * hdp)
    for lyr in agg.transformer.decoder.layers:
        layers.append({
            "wqkv": c(lyr.self_attn.in_proj_weight), "bqkv": c(lyr.self_attn.in_proj_bias),
            "wo": c(lyr.self_attn.out_proj.weight), "bo": c(lyr.self_attn.out_proj.bias),
            "cab": c(lyr.multihead_attn.out_proj.bias),
            "ln1g": c(lyr.norm1.weight), "ln1b": c(lyr.norm1.bias),
            "ln2g": c(lyr.norm2.weight), "ln2b": c(lyr.norm2.bias),
            "ln3g": c(lyr.norm3.weight), "ln3b": c(lyr.norm3.bias),
            "w1": c(lyr.linear1.weight), "b1": c(lyr.linear1.bias),
            "w2": c(lyr.linear2.weight), "b2": c(lyr.linear2.bias),
            "eps": float(lyr.norm1.eps),
        })
        if pad is not None:
            row, col, di = pad
            lay = layers[-1]
            wq = torch.zeros((3 * di, d), device=row.device, dtype=torch.float32)
            wq[row] = lay["wqkv"]
            bq = torch.zeros((3 * di,), device=row.device, dtype=torch.float32)
            bq[row] = lay["bqkv"]
            wo = torch.zeros((d, di), device=row.device, dtype=torch.float32)
            wo[:, col] = lay["wo"]
            lay.update({"wqkv": wq, "bqkv": bq, "wo": wo, "_unpad": (row, col)})
    return layers


def pack_aggregator(agg) -> Dict[str, object]:
    """The aggregator's own tensors for a STANDALONE ``TransformerAggregator.forward`` (reference model/aggregator.py:58-76): the
    decoder layers, the final norm, the special token and proj_in; the classifier slot of the token-0 tail is a zero [1, d] row
    (its logit is discarded).  Cached on the module, invalidated by parameter version counters like :func:`pack_level`."""
    params = getattr(agg, "_paths_param_list", None)
    if params is None:
        params = [agg.proj_in.weight, agg.proj_in.bias, agg.special_token] + list(agg.transformer.decoder.parameters())
        object.__setattr__(agg, "_paths_param_list", params)
    key = _versions(params)
    cache = getattr(agg, "_paths_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        c = lambda t: t.detach().float().contiguous()
        d = agg.dim
        dev = agg.proj_in.weight.device
        div_1d, div_2d = _pe_divs(d, dev)
        packed = {"layers": _pack_decoder_layers(agg), "lnfg": c(agg.transformer.decoder.norm.weight), "lnfb": c(agg.transformer.decoder.norm.bias),
                  "lnf_eps": float(agg.transformer.decoder.norm.eps), "special": c(agg.special_token), "bp": c(agg.proj_in.bias),
                  "wp": c(agg.proj_in.weight), "div_1d": div_1d, "div_2d": div_2d,
                  "wcls": torch.zeros((1, d), device=dev, dtype=torch.float32), "bcls": torch.zeros((1,), device=dev, dtype=torch.float32),
                  "_owner": agg}
    agg._paths_pack = (key, packed)
    return packed


def fast_path(mc) -> bool:
    """The shipped aggregator geometry (trans_dim 128, 4 heads, importance hidden 128): specialised, tuned kernels.  Anything
    else the reference's config surface allows runs on the shape-generic kernels (csrc/generic.hip + the f32 GEMM)."""
    return mc.trans_dim == 128 and mc.trans_heads == 4 and mc.importance_mlp_hidden_dim == 128


GENERIC_ADD = os.environ.get("PATHS_GENERIC_ADD", "1") != "0"       # other geometries: tuned LSTM kernels, importance / proj on x + h1 in flight
WS_CHAIN_192 = os.environ.get("PATHS_WS_CHAIN_192", "1") != "0"     # trans_dim 192: full layers' row chain on tlayer_ws_kernel<192>
WS_IMAGES_192 = os.environ.get("PATHS_WS_IMAGES_192", "1") != "0"   # ... and the first in_proj straight into the attention's head_dim-48 operand images
TAIL_WS_192 = os.environ.get("PATHS_TAIL_WS_192", "1") != "0"       # ... and the last layer at token 0 + head in ONE launch (token0_dist_kernel<192>)
FUSE_IMPORTANCE_TOKENS = os.environ.get("PATHS_FUSE_IMPORTANCE_TOKENS", "1") != "0"   # generic geometries: importance + token rows in one launch, PE from the table


def padded_head_dim(hd: int) -> int:
    """The head width the attention kernels run for a reference head_dim ``hd`` (reference model/aggregator.py:25-33 accepts any
    trans_dim % trans_heads == 0): 16 / 32 / 48 / 64 up to 64, multiples of 32 above.  A head that is narrower than that is ZERO-PADDED
    (rows of in_proj, columns of out_proj: :func:`_pack_decoder_layers`): padded q / k dims add nothing to a score, padded v dims yield
    zeros that meet zero columns of out_proj - the same function, evaluated on the supported width (softmax scale from the TRUE width)."""
    if hd <= 64:
        return next(w for w in (16, 32, 48, 64) if w >= hd)
    return (hd + 31) // 32 * 32


def wide_head(hd: int) -> bool:
    """head_dim above the flash-style kernels' 64: the three-step form of csrc/attn_wide.hip (score matrix in scratch)."""
    return hd > 64 and hd % 32 == 0 and hd <= 1024


def check_aggregator_geometry(d: int, H: int):
    """(trans_dim, trans_heads) pairs the aggregator kernels run (standalone ``TransformerAggregator.forward``)."""
    if d % 32 or d > 2048 or H < 1 or d % H or padded_head_dim(d // H) > 1024 or H * padded_head_dim(d // H) > 2048:
        raise NotImplementedError("the aggregator kernels need trans_dim % 32 == 0 (<= 2048, also after padding its heads to 16 / 32 / 48 / 64 "
                                  f"or a multiple of 32) and trans_dim % trans_heads == 0 (got trans_dim {d}, trans_heads {H})")


def check_supported(mc, training: bool = False):
    """Configurations this build runs on the HIP path; everything else is rejected loudly."""
    if not fast_path(mc):
        d, H, Hi = mc.trans_dim, mc.trans_heads, mc.importance_mlp_hidden_dim
        if d % 32 or d > 2048 or H < 1 or d % H or padded_head_dim(d // H) > 1024 or H * padded_head_dim(d // H) > 2048 or Hi < 1 or Hi > 1024:
            raise NotImplementedError("the shape-generic aggregator kernels need trans_dim % 32 == 0 (<= 2048, also after padding its heads to 16 / 32 / 48 / 64 "
                                      "or a multiple of 32), trans_dim % trans_heads == 0 and importance_mlp_hidden_dim <= 1024 "
                                      f"(got trans_dim {d}, trans_heads {H}, importance hidden {Hi})")
        if training and Hi % 4:
            raise NotImplementedError("training at aggregator geometries other than trans_dim=128 / 4 heads / importance hidden 128 needs "
                                      f"importance_mlp_hidden_dim % 4 == 0 (got {Hi}); inference runs")
    if mc.patch_embed_dim % 128 or mc.hierarchical_ctx_mlp_hidden_dim % 64:
        raise NotImplementedError("patch_embed_dim must be a multiple of 128 and hierarchical_ctx_mlp_hidden_dim of 64")
    if mc.pos_encoding_mode not in ("1d", "2d"):
        # the reference itself fails for any other value (size mismatch at the special-token concat)
        raise RuntimeError(f"pos_encoding_mode '{mc.pos_encoding_mode}' skips proj_in in the reference and fails there too")
    if mc.slide_ctx_mode not in ("residual", "concat", "none") or mc.importance_mode not in ("mul", "none"):
        raise ValueError("unknown slide_ctx_mode / importance_mode")


def _pad_rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    """[N, K] -> [ceil(N / mult) * mult, K] with zero rows (the f32 GEMM kernels read whole 128-row weight tiles)."""
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return w.contiguous()
    out = torch.zeros((n_pad, w.shape[1]), device=w.device, dtype=w.dtype)
    out[:n] = w
    return out


def generic_pack(lvl_pack: Dict[str, object], mc) -> Dict[str, object]:
    """Zero-padded fp32 weight copies for the generic path (cached in the level's pack dict, rebuilt with it)."""
    if "generic" not in lvl_pack:
        Hi = mc.importance_mlp_hidden_dim
        g = {"layers": []}
        if "w_ip" in lvl_pack:            # (absent in the pack of a standalone TransformerAggregator: pack_aggregator)
            g.update({"w1": _pad_rows(lvl_pack["w_ip"][:Hi]), "wp": _pad_rows(lvl_pack["w_ip"][Hi:])})
        for lay in lvl_pack["layers"]:
            g["layers"].append({k: _pad_rows(lay[k]) for k in ("wqkv", "wo", "w1", "w2")})
        lvl_pack["generic"] = g
    return lvl_pack["generic"]


GENERIC_SPLIT = os.environ.get("PATHS_GENERIC_SPLIT", "1") != "0"   # shape-generic INFERENCE: big GEMMs on the split-fp16 matrix-core kernel


def gemm_f32(a, lda: int, w_pad: torch.Tensor, bias, out, ldo: int, M: int, N: int, K: int, act: int = 0, residual=None, ldr: int = 0,
             split=None):
    """out[M, N] = act(a[M, K] w^T + bias) (+ residual) on the f32-input matrix cores (exact fp32 FMA chains).
    ``split`` = (cache dict, key) (inference only: the image costs one host sync per weight version): products with M >= 1024 rows run
    on the split-operand kernel of the tuned path instead (csrc/gemm_x6.hip: two fp16 planes per operand in the default mode, 22-bit
    products, fp32 accumulate - the arithmetic of every big product of the shipped geometry), 3x the f32-MFMA rate."""
    ptr = lambda t: t if isinstance(t, int) or t is None else t.data_ptr()
    if split is not None and GENERIC_SPLIT and GEMM_MODE != "f32" and M >= 1024 and K >= 128 and K % 32 == 0:
        cache, key = split
        k6 = f"{key}_x6_{split_planes()}"
        if k6 not in cache:
            cache[k6] = x6_pack(w_pad[:N].contiguous(), n_pad=(N + 255) // 256 * 256)
        img, ws = cache[k6]
        _lib.call("paths_gemm_nt_x6", ptr(a), lda, img.data_ptr(), K, 0, ptr(bias), ptr(out), ldo, M, N, (N + 255) // 256 * 256, K, act,
                  ptr(residual), ldr, None, 0, 0, split_planes(), ws, a_scale(), _lib.stream())
        return
    _lib.call("paths_gemm_nt_f32", ptr(a), lda, ptr(w_pad), K, ptr(bias), ptr(out), ldo, M, N, w_pad.shape[0], K, act,
              ptr(residual), ldr, None, 0, 0, _lib.stream())


def importance_proj_generic(mc, lvl_pack, src, ld_src: int, locs, num_ims, B: int, N: int, D: int, imp_mul: int, imp_out, tokens):
    """importance MLP + masked sigmoid + alpha * proj_in + positional encoding + special token for any (trans_dim, hidden) widths
    (reference model/paths.py:95-98,119-124; model/aggregator.py:37-65): two GEMMs and two row kernels."""
    gp = generic_pack(lvl_pack, mc)
    d, Hi, M = mc.trans_dim, mc.importance_mlp_hidden_dim, B * N
    dev = locs.device
    st = _lib.stream()
    p = _lib.ptr
    hid = torch.empty((M, Hi), device=dev, dtype=torch.float32)
    gemm_f32(src, ld_src, gp["w1"], lvl_pack["b1"], hid, Hi, M, Hi, D, act=1, split=(gp, "w1"))
    _lib.call("paths_importance_rows", p(hid), Hi, p(lvl_pack["w2"]), p(lvl_pack["b2"]), p(num_ims), N, M, Hi, p(imp_out), 0, st)
    proj = torch.empty((M, d), device=dev, dtype=torch.float32)
    gemm_f32(src, ld_src, gp["wp"], None, proj, d, M, d, D, split=(gp, "wp"))
    pe_mode = 2 if mc.pos_encoding_mode == "2d" else 1
    _lib.call("paths_tokens_assemble", p(proj), d, p(imp_out), imp_mul, p(lvl_pack["bp"]), p(lvl_pack["special"]),
              p(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), p(locs), N, mc.patch_size, pe_mode, d, B, p(tokens), st)


def importance_proj_generic_add(mc, lvl_pack, src, x_rows, add, locs, num_ims, B: int, N: int, D: int, imp_mul: int, imp_out, tokens, skip_padding: bool,
                                pe_tab=None):
    """:func:`importance_proj_generic` on the split-operand kernel with the GEMM input ``src + add`` summed while it is staged (the
    Y = X + h1 form of the tuned path: Y is never stored, ``src`` may be row addresses into the resident grids): the LSTM part of the
    selection chain does not depend on the aggregator's geometry, so every geometry gets the tuned gate kernels."""
    gp = generic_pack(lvl_pack, mc)
    d, Hi, M = mc.trans_dim, mc.importance_mlp_hidden_dim, B * N
    dev = locs.device
    st = _lib.stream()
    p = _lib.ptr
    nim = p(num_ims) if skip_padding else None

    # ONE product for the importance hidden layer and the projection (round 5): [W1 ; Wp] as one [Hi + d, D] weight - two launches of
    # M / 128 blocks each filled half the chip one after the other (2 x 56 us on the critical path at trans_dim 192), one launch of
    # twice the column tiles fills it once.  The relu of the hidden layer moves into paths_importance_rows (relu = 1).
    n = Hi + d
    n_pad = min((n + 255) // 256 * 256, (n + 191) // 192 * 192)      # (128 x 256 or 128 x 192 tiles, whichever pads less: 320 -> 384, not 512)
    k6 = f"w1p_x6_{split_planes()}"
    if k6 not in gp:
        gp["w1p"] = torch.cat([gp["w1"][:Hi], gp["wp"][:d]], dim=0).contiguous()
        gp["b1p"] = torch.cat([lvl_pack["b1"], torch.zeros((d,), device=dev, dtype=torch.float32)])
        gp[k6] = x6_pack(gp["w1p"], n_pad=n_pad)
    img, ws = gp[k6]
    # (skipped tiles of padding stay undefined: paths_importance_rows writes 0 for padded rows without looking at them and
    # paths_tokens_assemble selects on that 0 - only when the importance does not multiply the tokens does the buffer need a fill;
    # _lib.zeros: the fill is repeated when a recorded launch tape is replayed)
    hp_ = (_lib.zeros if (skip_padding and not imp_mul) else torch.empty)((M, n), device=dev, dtype=torch.float32)
    _lib.call("paths_gemm_add_nt_x6", p(src) if x_rows is None else None, D, p(x_rows), p(add), add.stride(1), img.data_ptr(), D, p(gp["b1p"]), p(hp_), n,
              M, n, n_pad, D, 0, nim, N, ws, a_scale(), st)
    pe_mode = 2 if mc.pos_encoding_mode == "2d" else 1
    if pe_tab is not None and FUSE_IMPORTANCE_TOKENS:
        # importance + tokens in one pass over the product's rows, sin / cos from the table (csrc/generic.hip: 6.5 + 13.8 us -> one launch)
        _lib.call("paths_importance_tokens_rows", p(hp_), n, p(lvl_pack["w2"]), p(lvl_pack["b2"]), p(num_ims), N, M, Hi, p(imp_out), 1, imp_mul,
                  p(lvl_pack["bp"]), p(lvl_pack["special"]), p(pe_tab), pe_tab.shape[0], p(locs), mc.patch_size, pe_mode, d, p(tokens), st)
        return
    _lib.call("paths_importance_rows", p(hp_), n, p(lvl_pack["w2"]), p(lvl_pack["b2"]), p(num_ims), N, M, Hi, p(imp_out), 1, st)
    _lib.call("paths_tokens_assemble", hp_.data_ptr() + 4 * Hi, n, p(imp_out), imp_mul, p(lvl_pack["bp"]), p(lvl_pack["special"]),
              p(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), p(locs), N, mc.patch_size, pe_mode, d, B, p(tokens), st)


def fp8_pack(lvl_pack: Dict[str, object], mc) -> Dict[str, object]:
    """e4m3 images + per-tensor scales (device scalars) of the aggregator's four weight matrices per layer (cached in the level's pack
    dict, rebuilt with it), plus the scratch words the scale kernels use."""
    if "fp8" not in lvl_pack:
        dev = lvl_pack["wcls"].device
        st = _lib.stream()
        scratch = torch.zeros((1,), device=dev, dtype=torch.int32)
        g = {"layers": [], "scratch": scratch, "a_scale": torch.ones((1,), device=dev, dtype=torch.float32)}
        for lay in lvl_pack["layers"]:
            e = {}
            for k in ("wqkv", "wo", "w1", "w2"):
                w = lay[k].contiguous()
                N, K = w.shape
                w8 = torch.empty(((N + 255) // 256 * 256, K), device=dev, dtype=torch.uint8)
                sc = torch.empty((1,), device=dev, dtype=torch.float32)
                _lib.call("paths_fp8_pack_weight", w.data_ptr(), K, N, K, w8.data_ptr(), sc.data_ptr(), scratch.data_ptr(), st)
                e[k] = (w8, sc)
            g["layers"].append(e)
        lvl_pack["fp8"] = g
    return lvl_pack["fp8"]


def _fp8_image(fp, key: str, nbytes: int, dev, zero: bool = False) -> torch.Tensor:
    buf = fp.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = fp[key] = (torch.zeros if zero else torch.empty)((nbytes,), device=dev, dtype=torch.uint8)
    return buf


def gemm_fp8(fp, a, lda: int, w8sc, bias, out, ldo: int, M: int, N: int, K: int, act: int = 0, residual=None, ldr: int = 0, tok=None):
    """out[M, N] = act(a[M, K] w^T + bias) (+ residual) with e4m3 operands (csrc/gemm_fp8.hip): per-tensor activation scale from a
    device-side max|a| (no host sync), the activation image from one quantisation pass, the weight image and its scale from :func:`fp8_pack`."""
    ptr = lambda t: t if isinstance(t, int) or t is None else t.data_ptr()
    st = _lib.stream()
    # tok = (num_ims, T): the rows are tokens of slides - only valid ones count for the scale (padded rows may hold anything)
    _lib.call("paths_fp8_scale", ptr(a), lda, M, K, fp["a_scale"].data_ptr(), fp["scratch"].data_ptr(), tok[0].data_ptr() if tok else None,
              tok[1] if tok else 0, st)
    a8 = _fp8_image(fp, "a8", (M + 255) // 256 * 256 * K, w8sc[0].device)
    _lib.call("paths_fp8_quantize", ptr(a), lda, M, K, fp["a_scale"].data_ptr(), a8.data_ptr(), st)
    _lib.call("paths_gemm_nt_fp8", a8.data_ptr(), w8sc[0].data_ptr(), fp["a_scale"].data_ptr(), w8sc[1].data_ptr(), ptr(bias), ptr(out), ldo,
              M, N, K, act, ptr(residual), ldr, st)


def ffn_fp8(fp, lay8, x2, d: int, b1, b2, y2, M: int, tok=None):
    """y2 = x2 + linear2(relu(linear1(x2))) with e4m3 operands and the hidden layer [M, 4d] handed from the first GEMM to the second
    as an e4m3 image (never as fp32: 1.6 GB written, read twice for max|.| and quantisation, at the stress shape).  Its per-tensor
    scale is CALIBRATED: the first call of a layer runs the fp32 hand-over and records 448 / (2 max|hidden|) (a factor 2 of head
    room; values beyond saturate), later calls reuse it; max|hidden| of every call lands in lay8["h_amax"] (float bits) for checks."""
    F = 4 * d
    dev = x2.device
    st = _lib.stream()
    if "h_scale" not in lay8:
        hff = torch.empty((M, F), device=dev, dtype=torch.float32)
        gemm_fp8(fp, x2, d, lay8["w1"], b1, hff, F, M, F, d, act=1, tok=tok)
        sc = torch.empty((1,), device=dev, dtype=torch.float32)
        _lib.call("paths_fp8_scale", hff.data_ptr(), F, M, F, sc.data_ptr(), fp["scratch"].data_ptr(), tok[0].data_ptr() if tok else None,
                  tok[1] if tok else 0, st)
        sc *= 0.5
        lay8["h_scale"] = sc
        lay8["h_amax"] = torch.zeros((1,), device=dev, dtype=torch.int32)
        gemm_fp8(fp, hff, F, lay8["w2"], b2, y2, d, M, d, F, residual=x2, ldr=d, tok=tok)
        return
    _lib.call("paths_fp8_scale", x2.data_ptr(), d, M, d, fp["a_scale"].data_ptr(), fp["scratch"].data_ptr(), tok[0].data_ptr() if tok else None,
              tok[1] if tok else 0, st)
    a8 = _fp8_image(fp, "a8", (M + 255) // 256 * 256 * d, dev)
    _lib.call("paths_fp8_quantize", x2.data_ptr(), d, M, d, fp["a_scale"].data_ptr(), a8.data_ptr(), st)
    h8 = _fp8_image(fp, "h8", (M + 255) // 256 * 256 * F, dev, zero=True)
    lay8["h_amax"].zero_()
    _lib.call("paths_gemm_nt_fp8_out8", a8.data_ptr(), lay8["w1"][0].data_ptr(), fp["a_scale"].data_ptr(), lay8["w1"][1].data_ptr(), b1.data_ptr(),
              h8.data_ptr(), lay8["h_scale"].data_ptr(), lay8["h_amax"].data_ptr(), M, F, d, 1, st)
    _lib.call("paths_gemm_nt_fp8", h8.data_ptr(), lay8["w2"][0].data_ptr(), lay8["h_scale"].data_ptr(), lay8["w2"][1].data_ptr(), b2.data_ptr(),
              y2.data_ptr(), d, M, d, F, 0, x2.data_ptr(), d, st)


FP8_HEAD_DIMS = (32, 64, 128, 256, 384)      # csrc/attn_fp8.hip (round 5: the wide ones - 1536 / 4 heads = 384 is the stress row's own form)


def fp8_supported(mc) -> bool:
    d, H = mc.trans_dim, mc.trans_heads
    return d % 128 == 0 and d % H == 0 and (d // H) in FP8_HEAD_DIMS      # paths_gemm_nt_fp8 needs K % 128 == 0 (one 64-k instruction pair per stage)


def _aggregator_forward_generic(mc, lvl_pack, tokens, num_ims, ctx_prev, ctx_all, fp8: bool = False, status=None) -> Dict[str, torch.Tensor]:
    """The aggregator for any (trans_dim, heads): generic GEMMs + csrc/generic.hip (reference model/aggregator.py:58-76 with torch's
    post-LN decoder layers, model/paths.py:130-139).  The last layer is evaluated at token 0 only (its other rows are never read).
    ``fp8`` (ops.AGG_FP8, the BASELINE configs[4] stress variant, NOT a parity path): the products over all tokens - in_proj, the full
    layers' attention, out_proj and both feed-forward GEMMs - run with e4m3 operands (csrc/gemm_fp8.hip, csrc/attn_fp8.hip); LayerNorm,
    residuals, the last layer's token-0 attention and row chain and the classifier stay in fp32."""
    gp = generic_pack(lvl_pack, mc)
    B, T, d = tokens.shape
    H, L = mc.trans_heads, mc.trans_layers
    hd_true = d // H
    hd = padded_head_dim(hd_true)         # the head width the kernels run (narrower heads are zero-padded in the pack: padded_head_dim)
    di = H * hd                           # inner width of the attention: q | k | v are [M, 3 di], the attention output [B, T, di]
    dev = tokens.device
    st = _lib.stream()
    p = _lib.ptr
    f32 = dict(device=dev, dtype=torch.float32)
    qscale = LOG2E / math.sqrt(hd_true)
    M = B * T
    if fp8 and not fp8_supported(mc):
        raise NotImplementedError(f"the e4m3 aggregator needs trans_dim % 128 == 0 and head_dim in {FP8_HEAD_DIMS} (got {d} / {H} heads)")
    fp = fp8_pack(lvl_pack, mc) if fp8 else None
    ws8 = torch.empty((int(_lib.load().paths_attention_fp8_workspace(B, T, H, hd)),), device=dev, dtype=torch.uint8) if (fp8 and L > 1) else None
    # full layers' attention on the split-fp16 matrix-core kernel for any head_dim (the arithmetic of the tuned path), unless the
    # f32 mode is selected; the last layer's single query stays on the f32-input kernel
    wide = wide_head(hd)
    wsw = torch.empty((int(_lib.load().paths_attention_wide_workspace(T, hd)),), **f32) if wide else None
    h3 = (not fp8) and GENERIC_SPLIT and GEMM_MODE == "h3" and L > 1 and not wide
    wsh = torch.empty((int(_lib.load().paths_attention_h3_any_workspace(B, T, H, hd)),), device=dev, dtype=torch.uint8) if h3 else None
    x = tokens.view(M, d)
    qkv = torch.empty((M + (128 if wide else 0), 3 * di), **f32)     # (wide heads: the score product reads whole 128-row tiles of k)
    if wide:
        qkv[M:].zero_()
    # rows of padded queries are never written by the attention kernels: harmless row-wise garbage on the accurate path, but the e4m3
    # path takes max|.| over WHOLE activation matrices for its per-tensor scales - there they must be defined (zero)
    attn = (torch.zeros if fp8 else torch.empty)((B, T, di), **f32)
    rows, ldx = M, d                      # the current activation: `rows` rows, row stride ldx (the last layer keeps token 0 of every slide)
    lda_attn = di                         # row stride of the attention output as the out_proj operand (T * di for the last layer's token-0 rows)
    ws192 = WS_CHAIN_192 and d == 192 and hd == hd_true and not fp8 and GEMM_MODE == "h3" and GENERIC_SPLIT and not wide
    # ... and the LAST layer + decoder.norm + context + classifier in the one launch of the token-0 tail (csrc/token0_ws.hip at 192:
    # the K / V projections folded into the query, so the chain launch in front of it stops at the layer's input rows)
    tail192 = (ws192 and TAIL_WS and TAIL_WS_192 and H == 4 and L >= 2 and h3
               and bool(_lib.load().paths_token0_ws_supported(B, T, d, H)))
    qkv_ready = img_ready = False
    for l in range(L):
        lay, gl = lvl_pack["layers"][l], gp["layers"][l]
        last = l == L - 1
        big = fp8 and not last            # products over all tokens of a full layer

        def gemm(a, lda, key, bias, out, ldo, m, n, kdim, act=0, residual=None, ldr=0, low=False):
            if low:
                gemm_fp8(fp, a, lda, fp["layers"][l][key], bias, out, ldo, m, n, kdim, act, residual, ldr, tok=(num_ims, T) if m == M else None)
            else:
                gemm_f32(a, lda, gl[key], bias, out, ldo, m, n, kdim, act, residual, ldr, split=(gl, key))

        if qkv_ready:                         # the previous layer's chain launch already projected this layer's q | k | v
            qkv_ready = False
        elif ws192 and WS_IMAGES_192 and h3 and not last and H == 4:
            # q | k | v straight into the head_dim-48 operand images of the attention (no fp32 rows, no prep launch)
            iq, sq = tlayer_ws_images(lay, 1)
            _lib.call("paths_token_layer_ws", p(x), None, None, None, None, p(iq), None, None, None, None, None, None, None, None, None, None,
                      p(lay["bqkv"]), 1.0, 1.0, 1.0, sq[0], p(wsh), p(num_ims), B, T, d, H, 0, 1, 1, qscale, lay["eps"], None, 0, st)
            img_ready = True
        elif ws192:
            iq, sq = tlayer_ws_images(lay, 1)
            _lib.call("paths_token_layer_ws_rows", p(x), None, None, None, p(iq), None, None, None, None, None, None, None, None, None, None,
                      p(lay["bqkv"]), 1.0, 1.0, 1.0, sq[0], p(qkv), 3 * d, p(num_ims), B, T, d, 0, 1, 1, lay["eps"], st)
        else:
            gemm(x, d, "wqkv", lay["bqkv"], qkv, 3 * di, M, 3 * di, d, low=fp8)        # (the last layer's K / V cover all tokens too)
        if big:
            _lib.call("paths_attention_fp8_qkv", p(qkv), 3 * di, p(attn), p(num_ims), B, T, H, hd, qscale, p(ws8), st)
        elif wide:
            _lib.call("paths_attention_wide_fwd", p(qkv), 3 * di, p(attn), None, p(num_ims), B, T, H, hd, qscale, 1 if last else 0, 0, 0.0, p(wsw), st)
        elif h3 and not last and img_ready:
            _lib.call("paths_attention_h3_any_img", p(attn), p(num_ims), B, T, H, hd, p(wsh), st)
            img_ready = False
        elif h3 and not last:
            _lib.call("paths_attention_h3_any", p(qkv), 3 * di, p(attn), p(num_ims), B, T, H, hd, qscale, p(wsh), st)
        elif last and GENERIC_SPLIT:
            # single query per (slide, head), keys split over workgroups (csrc/attn_token0.hip): [B, d] instead of row 0 of [B, T, d]
            a0 = torch.empty((B, di), **f32)
            ws0 = torch.empty((int(_lib.load().paths_attention_token0_workspace(B, T, H)),), **f32)
            _lib.call("paths_attention_token0_any", p(qkv), 3 * di, p(num_ims), p(a0), p(ws0), B, T, H, hd, qscale, st)
        else:
            _lib.call("paths_attention_any", p(qkv), 3 * di, p(attn), p(num_ims), B, T, H, hd, qscale, 1 if last else 0, st)
        if ws192 and not last:
            # the reference's dataclass-default width (config.py:30): out_proj + norm1 + cross-attention bias + norm2 + feed-forward +
            # norm3 of a full layer AND the next layer's in_proj in ONE launch of the weight-stationary chain kernel (csrc/tlayer_ws.hip
            # instantiated at 192; attention output in, q | k | v out as fp32 rows) instead of four GEMMs and two LayerNorm launches
            nxt = lvl_pack["layers"][l + 1]
            ip, sp = tlayer_ws_images(lay, 0)
            x3 = torch.empty((M, d), **f32)
            if tail192 and l + 1 == L - 1:
                _lib.call("paths_token_layer_ws_rows", p(x), p(attn), p(x3), p(ip), None, p(lay["bo"]), p(lay["ln1g"]), p(lay["ln1b"]), p(lay["cab"]),
                          p(lay["ln2g"]), p(lay["ln2b"]), p(lay["b1"]), p(lay["b2"]), p(lay["ln3g"]), p(lay["ln3b"]), None, sp[0], sp[1], sp[2],
                          1.0, None, 0, p(num_ims), B, T, d, 1, 0, 1, lay["eps"], st)
                x = x3
                break
            iq, sq = tlayer_ws_images(nxt, 1)
            qkv_next = torch.empty((M, 3 * d), **f32)
            _lib.call("paths_token_layer_ws_rows", p(x), p(attn), p(x3), p(ip), p(iq), p(lay["bo"]), p(lay["ln1g"]), p(lay["ln1b"]), p(lay["cab"]),
                      p(lay["ln2g"]), p(lay["ln2b"]), p(lay["b1"]), p(lay["b2"]), p(lay["ln3g"]), p(lay["ln3b"]), p(nxt["bqkv"]), sp[0], sp[1], sp[2],
                      sq[0], p(qkv_next), 3 * d, p(num_ims), B, T, d, 1, 1, 1, lay["eps"], st)
            x, ldx, qkv, qkv_ready = x3, d, qkv_next, True
            continue
        if last:
            rows, ldx, lda_attn = B, T * d, T * di    # rows = token 0 of every slide: row stride T * d (T * di) into the [B, T, .] tensors
        y1 = torch.empty((rows, d), **f32)
        if last and GENERIC_SPLIT and not wide:
            gemm(a0, di, "wo", lay["bo"], y1, d, rows, d, di, residual=x, ldr=ldx)
        else:
            gemm(attn, lda_attn, "wo", lay["bo"], y1, d, rows, d, di, residual=x, ldr=ldx, low=big)
        x1 = torch.empty((rows, d), **f32)
        x2 = y1                           # (in place: every wave reads its row before it writes it)
        _lib.call("paths_layernorm2_rows", p(y1), d, p(lay["ln1g"]), p(lay["ln1b"]), p(lay["cab"]), p(lay["ln2g"]), p(lay["ln2b"]), p(x2), d,
                  rows, d, lay["eps"], st)
        y2 = x1                           # (re-used)
        if big:
            ffn_fp8(fp, fp["layers"][l], x2, d, lay["b1"], lay["b2"], y2, rows, tok=(num_ims, T))
        else:
            hff = torch.empty((rows, 4 * d), **f32)
            gemm(x2, d, "w1", lay["b1"], hff, 4 * d, rows, 4 * d, d, act=1)
            gemm(hff, 4 * d, "w2", lay["b2"], y2, d, rows, d, 4 * d, residual=x2, ldr=d)
        x3 = torch.empty((rows, d), **f32)
        _lib.call("paths_layernorm_rows", p(y2), d, None, p(lay["ln3g"]), p(lay["ln3b"]), p(x3), d, rows, d, lay["eps"], st)
        x, ldx = x3, d
    nlog = lvl_pack["wcls"].shape[0]
    ctx_out = torch.empty((B, d), **f32)
    logits = torch.empty((B, nlog), **f32)
    res = ctx_prev if mc.slide_ctx_mode == "residual" else None
    cat = ctx_all.contiguous() if (mc.slide_ctx_mode == "concat" and ctx_all is not None and ctx_all.shape[1] > 0) else None
    if tail192:
        # x: the last layer's input rows [B, T, d] (special token first: the reference's order)
        w = lvl_pack["layers"][L - 1]
        img = token0_ws_image(w, qscale)
        part = torch.empty((int(_lib.load().paths_token0_ws_partials_d(B, T, d)),), **f32)
        cnt = token0_counters(dev, B)
        _lib.call(
            "paths_token0_tail_ws", p(x), p(num_ims), p(img), w["bqkv"].data_ptr() + 4 * 2 * d, p(w["bo"]), p(w["ln1g"]), p(w["ln1b"]),
            p(w["cab"]), p(w["ln2g"]), p(w["ln2b"]), p(w["b1"]), p(w["b2"]), p(w["ln3g"]), p(w["ln3b"]),
            p(lvl_pack["lnfg"]), p(lvl_pack["lnfb"]), p(res) if res is not None else None,
            res.stride(0) if res is not None else 0, p(cat) if cat is not None else None, cat.shape[1] if cat is not None else 0,
            p(lvl_pack["wcls"]), p(lvl_pack["bcls"]), nlog, lvl_pack["wcls"].shape[1], p(ctx_out), p(logits),
            p(part), p(cnt), p(status) if status is not None else None, B, T, d, H, w["eps"], lvl_pack["lnf_eps"], 0, st)
        return {"logits": logits, "ctx_slide": ctx_out}
    # x: [B, d] if the loop ended on the last layer's token-0 rows (L >= 1), row stride d
    _lib.call("paths_final_head_any", p(x), d if L >= 1 else T * d, p(lvl_pack["lnfg"]), p(lvl_pack["lnfb"]), p(res) if res is not None else None,
              res.stride(0) if res is not None else 0, p(cat) if cat is not None else None, cat.shape[1] if cat is not None else 0,
              p(lvl_pack["wcls"]), p(lvl_pack["bcls"]), nlog, lvl_pack["wcls"].shape[1], p(ctx_out), p(logits), B, d, lvl_pack["lnf_eps"], st)
    return {"logits": logits, "ctx_slide": ctx_out}


# ---------------------------------------------------------------------------------------------
# one level
# ---------------------------------------------------------------------------------------------
def level_forward(mc, lstm_pack, lvl_pack, fts: torch.Tensor, locs: torch.Tensor, num_ims: torch.Tensor,
                  state_prev: Optional[torch.Tensor], ctx_prev: Optional[torch.Tensor], ctx_all: Optional[torch.Tensor],
                  skip_padding: bool) -> Dict[str, torch.Tensor]:
    """fts [B,N,D] fp32 contiguous; locs [B,N,2] int64; num_ims [B] int64;
    state_prev: [B,N,>=D+Hc] view whose last dim holds (h|c) of the previous level (row stride arbitrary) or None;
    ctx_prev [B,d] (residual source) or None; ctx_all [B,depth,d] contiguous (concat mode) or None."""
    sel = selection_forward(mc, lstm_pack, lvl_pack, fts, locs, num_ims, state_prev, skip_padding)
    status = torch.zeros((1,), device=fts.device, dtype=torch.int32)
    agg = aggregator_forward(mc, lvl_pack, sel["tokens"], sel["num_ims"], ctx_prev, ctx_all, status=status, qkv=sel)
    # the drop-in call is synchronous anyway (the range guard above it syncs): a token-0 tail whose bounded hand-off wait gave up
    # (status bit 2, csrc/token0_ws.hip) must not hand back its logits
    if int(status.item()) & 4:
        raise _lib.PathsHipError("a bounded in-launch hand-off wait of the token-0 tail gave up (csrc/token0_ws.hip): results invalid")
    return {"logits": agg["logits"], "ctx_slide": agg["ctx_slide"], "ctx_patch": sel["ctx_patch"], "importance": sel["importance"]}


def selection_forward(mc, lstm_pack, lvl_pack, fts, locs, num_ims, state_prev, skip_padding: bool,
                      parent=None, max_pos: int = 0, x_rows=None, feat_dim: Optional[int] = None,
                      importance_out: Optional[torch.Tensor] = None, last_level: bool = False,
                      topk: Optional[Dict[str, object]] = None) -> Dict[str, torch.Tensor]:
    """The part of a level that decides the NEXT level: LSTM state update, importance, token projection
    (reference model/paths.py:71-124).  Returns ctx_patch (new state), importance, tokens, num_ims.

    ``last_level`` (device recursion): no top-K follows, so nothing waits for the importance alone - the fused finish then runs as ONE
    launch (FUSE_QKV mode 1) instead of an importance-only finish plus the tokens / images finish.

    ``topk`` (device recursion, optional) = {"keep": int, "zero_row": tensor, "status": tensor}: the caller will select the ``keep`` most
    important patches of every slide next.  When the fused finish runs in mode 2 the selection happens INSIDE the importance finish
    (FUSE_TOPK) and the result comes back as "keep_idx" [B, cap] int32, "keep_count" [B] int32 and "kept_rows" [B, cap] int64 (addresses
    of the kept patches' h rows in ctx_patch) - exactly what paths_topk_rows would write; absent keys = the caller runs the top-K itself.

    ``max_pos`` (optional): an upper bound (exclusive) of ``locs // patch_size`` known to the caller (the grid size of the level);
    positional-encoding values are then read from a cached table instead of evaluated per token element.

    ``x_rows`` (device recursion only, default split mode): [B, N] int64 ADDRESSES of the feature rows (paths_gather_rows
    ``row_ptrs``); ``fts`` is then None, ``feat_dim`` = D, and the GEMMs read the rows where they live (no gathered copy).

    ``parent`` (device recursion only) = {"hp": [rows, 3Hc+D], "hp_row": [B,N] int32, "c0": [B,N,Hc]}: the up-to-4 children
    of a kept patch share the parent's h, so the h half of the gate GEMM is computed once per kept PARENT
    (:func:`parent_partials`) and added in the children's epilogue; ``state_prev`` is then None."""
    _lib.require_cuda(fts, locs, num_ims, state_prev, x_rows)
    if x_rows is not None:
        assert fts is None and feat_dim is not None and state_prev is None and x_rows.dtype == torch.int64 and x_rows.is_contiguous()
        (B, N), D = x_rows.shape, int(feat_dim)
    else:
        B, N, D = fts.shape
    d, H, L = mc.trans_dim, mc.trans_heads, mc.trans_layers
    T = N + 1
    M = B * N
    dev = locs.device
    st = _lib.stream()
    p = _lib.ptr
    f32 = dict(device=dev, dtype=torch.float32)
    assert fts is None or (fts.is_contiguous() and fts.dtype == torch.float32)
    locs = locs.contiguous()
    num_ims = num_ims.contiguous()
    assert locs.dtype == torch.int64 and num_ims.dtype == torch.int64
    nim = p(num_ims) if skip_padding else None
    pe_mode = 2 if mc.pos_encoding_mode == "2d" else 1
    tokens = torch.empty((B, T, d), **f32)

    x6 = use_x6(D, lstm_pack["Hc"] if mc.lstm else 64)
    assert x_rows is None or (x6 and split_planes() == 2 and mc.lstm), "row pointers need the default split mode and lstm=true"
    pe_rows = N if pe_mode == 1 else int(max_pos)
    if pe_rows == 0 and FUSE_QKV in (1, 2) and fast_path(mc) and N % 64 == 0 and not torch.is_grad_enabled():
        # a drop-in call (no grid size given): the fused finish reads its sin / cos values from the table only, so the table is sized
        # from the batch's largest position - one host sync, in a call that is synchronous anyway (level_forward reads the status word)
        top = int(locs.max().item()) // int(mc.patch_size) + 1 if locs.numel() else 0
        pe_rows = top if 0 < top <= (1 << 16) else 0
    pe_tab = pe_table(lvl_pack, pe_mode, d, pe_rows) if pe_rows > 0 else None

    generic = not fast_path(mc)
    # default inference form of the shipped geometry: in_proj of decoder layer 0 inside the importance / projection finish (FUSE_QKV)
    fuse_qkv = (FUSE_QKV in (1, 2) and not generic and x6 and split_planes() == 2 and mc.lstm and GEMM_MODE == "h3" and TLAYER_WS and QKV_IMAGES
                and L > 1 and pe_tab is not None and not (ATTN_FP8 or AGG_FP8) and D % 64 == 0 and D >= 256 and N % 64 == 0 and TAIL_WS)
    fused: Dict[str, object] = {}
    # any aggregator geometry on the tuned LSTM kernels: the importance / projection products take x + h1 summed while staged
    generic_add = generic and x6 and split_planes() == 2 and mc.lstm and GENERIC_ADD and GENERIC_SPLIT and D % 128 == 0

    def importance_proj(src, imp_mul, imp_out, add=None):
        """tokens / importance from ``src`` (+ ``add``: x6 only, the GEMM input is src + add, row stride of add arbitrary)."""
        if generic:
            assert add is None and src is not None
            return importance_proj_generic(mc, lvl_pack, src, D, locs, num_ims, B, N, D, imp_mul, imp_out, tokens)
        common = (p(lvl_pack["b1"]), p(lvl_pack["w2"]), p(lvl_pack["b2"]),
                  p(lvl_pack["bp"]), p(lvl_pack["special"]), p(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]),
                  p(pe_tab), pe_tab.shape[0] if pe_tab is not None else 0, p(locs),
                  p(num_ims), N, mc.patch_size, pe_mode, imp_mul, p(imp_out), p(tokens), None, None,
                  M, D, mc.importance_mlp_hidden_dim, d, 1 if skip_padding else 0, st)
        if x6:
            wip, wip_s = _x6_of(lvl_pack, "w_ip_fwd")
            splitk = SPLITK_IMPORTANCE and add is not None and split_planes() == 2 and (M + 127) // 128 <= 160
            if splitk and fuse_qkv and imp_out is importance:
                # GEMM rows in token order + a finish that also projects q | k | v of decoder layer 0 into the attention's operand images
                lay0 = lvl_pack["layers"][0]
                iq, sq = tlayer_ws_images(lay0, 1)
                hd = d // H
                ws = torch.empty((int(_lib.load().paths_importance_proj_x6_workspace(M)),), device=dev, dtype=torch.uint8)
                qkv_img = torch.empty((int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, 2)),), device=dev, dtype=torch.uint8)
                args = (p(src), D, p(x_rows) if src is None else None, p(add), add.stride(1), p(wip), p(lvl_pack["b1"]), p(lvl_pack["w2"]),
                        p(lvl_pack["b2"]), p(lvl_pack["bp"]), p(lvl_pack["special"]), p(pe_tab), pe_tab.shape[0], p(locs), p(num_ims), B, N,
                        mc.patch_size, pe_mode, imp_mul, p(imp_out), p(tokens), D, 1 if skip_padding else 0, wip_s, a_scale(), p(ws),
                        p(iq), p(lay0["bqkv"]), sq[0], LOG2E / math.sqrt(hd), p(qkv_img))
                # (token order of this form: patch i = token i, the special token at index num_ims[b]; the tail is told: special_last)
                no_topk = (0, None, 0, None, None, 0, None, None, None, None)
                if FUSE_QKV == 2 and not last_level:
                    if topk is not None and FUSE_TOPK and N <= 8192:
                        keep = int(topk["keep"])
                        cap = N if keep < 0 else min(N, keep)
                        fused["keep_idx"] = torch.empty((B, cap), device=dev, dtype=torch.int32)
                        fused["keep_count"] = torch.empty((B,), device=dev, dtype=torch.int32)
                        fused["kept_rows"] = torch.empty((B, cap), device=dev, dtype=torch.int64)
                        _lib.call("paths_importance_qkv_x6", *args, 9, 0, keep, p(fused["keep_idx"]), cap, p(fused["keep_count"]), p(state_out), Dp,
                                  p(fused["kept_rows"]), p(topk["zero_row"]), p(topk_counters(dev, B)), p(topk.get("status")), st)
                    else:
                        _lib.call("paths_importance_qkv_x6", *args, 3, 0, *no_topk, st)
                    # (the aggregator stream finishes the tokens: ws / qkv_img travel with the closure)
                    fused["finish"] = lambda: _lib.call("paths_importance_qkv_x6", *args, 4, 1, *no_topk, _lib.stream())
                else:
                    _lib.call("paths_importance_qkv_x6", *args, 5, 0, *no_topk, st)
                fused["qkv_img"], fused["ws"] = qkv_img, ws
                return
            # M/128 blocks fill half the chip at K = 2048 x 8 slides: two k halves on twice the blocks + an epilogue launch
            splitk_ws = None
            if splitk:
                splitk_ws = torch.empty((int(_lib.load().paths_importance_proj_x6_workspace(M)),), device=dev, dtype=torch.uint8)
            _lib.call("paths_importance_proj_x6", p(src), D, p(x_rows) if src is None else None, p(add),
                      add.stride(1) if add is not None else 0,
                      p(wip), *common[:-1], split_planes(), wip_s, a_scale(), p(splitk_ws), common[-1])
        else:
            assert add is None
            _lib.call("paths_importance_proj", p(src), D, p(lvl_pack["w_ip_fwd"]), *common)

    if importance_out is not None:          # caller's buffer, already zero where padding rows must read 0
        assert importance_out.shape == (B, N) and importance_out.is_contiguous() and importance_out.dtype == torch.float32
        importance = importance_out
    else:
        importance = torch.zeros((B, N), **f32) if skip_padding else torch.empty((B, N), **f32)
    if mc.lstm:
        Hc = lstm_pack["Hc"]
        Dp = D + Hc
        state_out = torch.empty((B, N, Dp), **f32)
        # x6: Y = X + h1 is summed inside the importance/proj GEMM's staging (the generic importance / projection GEMMs read a stored Y)
        y = None if (x6 and (not generic or generic_add)) else torch.empty((B, N, D), **f32)
        ws_o = torch.empty(((M + 255) // 256 * 256, D), **f32)       # gate scratch: whole 256-row tiles (raw accumulator layout)
        hp, hp_row = None, None
        if parent is not None:
            assert state_prev is None
            c0t = parent["c0"]
            assert c0t.shape == (B, N, Hc) and c0t.is_contiguous()
            ld, h0, c0 = Hc, None, c0t.data_ptr()
            hp, hp_row = p(parent["hp"]), p(parent["hp_row"])
        elif state_prev is not None:
            assert state_prev.shape[:2] == (B, N) and state_prev.shape[2] == Dp and state_prev.stride(2) == 1
            assert state_prev.stride(0) == N * state_prev.stride(1), "state rows must be uniformly strided"
            ld = state_prev.stride(1)
            h0, c0 = state_prev.data_ptr(), state_prev.data_ptr() + 4 * D
        else:
            ld, h0, c0 = 0, None, None

        def lstm(phases):
            if x6:
                (wg, wg_s), (wm, wm_s) = _x6_of(lstm_pack, "w_gates"), _x6_of(lstm_pack, "w_mem")
                _lib.call("paths_lstm_cell_x6", p(fts), D, p(x_rows), h0, ld, c0, ld, p(wg), p(lstm_pack["b_gates"]), p(wm), p(lstm_pack["b_mem"]),
                          p(state_out), Dp, p(y), D, p(ws_o), None, None, hp, hp_row, M, D, Hc, nim, N, phases,
                          split_planes(), wg_s, wm_s, a_scale(), st)
            else:
                _lib.call("paths_lstm_cell", p(fts), D, h0, ld, c0, ld, p(lstm_pack["w_gates"]), p(lstm_pack["b_gates"]),
                          p(lstm_pack["w_mem"]), p(lstm_pack["b_mem"]), p(state_out), Dp, p(y), D,
                          p(ws_o), None, None, hp, hp_row, M, D, Hc, nim, N, phases, st)

        if KERNEL_TIMER is None:
            lstm(7)
        else:                   # bench.py: bracket the dominant kernel (output-gate GEMM) with events on this stream
            timed("lstm_gate_c", lambda: lstm(1))
            timed("lstm_gate_o", lambda: lstm(2), {"rows": N, "B": B, "K": D if h0 is None else 2 * D, "Ncols": D, "parent_partials": parent is not None,
                                                   "x6": x6, "planes": split_planes() if x6 else 0}, detail=False)
            timed("lstm_mem_to_out", lambda: lstm(4))
        if generic_add:
            timed("importance_proj", lambda: importance_proj_generic_add(mc, lvl_pack, fts, x_rows, state_out, locs, num_ims, B, N, D,
                                                                         1 if mc.importance_mode == "mul" else 0, importance, tokens, skip_padding, pe_tab=pe_tab))
        elif x6 and not generic:
            timed("importance_proj", lambda: importance_proj(fts, 1 if mc.importance_mode == "mul" else 0, importance, add=state_out))
        else:
            timed("importance_proj", lambda: importance_proj(y, 1 if mc.importance_mode == "mul" else 0, importance))
        del ws_o
    else:
        # lstm=false (reference model/paths.py:95-109): alpha from X; Z = alpha*X (+ hctx_mlp(previous Z) on valid rows);
        # patch ctx = Z; tokens = proj_in(Z) + PE.  Re-uses the GEMM kernels; not a tuned path.
        importance_proj(fts, 0, importance)                      # pass 1: importance only (tokens overwritten below)
        hctx = None
        if state_prev is not None and mc.hierarchical_ctx:
            assert state_prev.shape == (B, N, D) and state_prev.stride(2) == 1 and state_prev.stride(0) == N * state_prev.stride(1)
            Hh = lvl_pack["wh1"].shape[0]
            assert Hh % 128 == 0, "hierarchical_ctx_mlp_hidden_dim must be a multiple of 128 for lstm=false"
            hid = torch.empty((B, N, Hh), **f32)
            hctx = torch.empty((B, N, D), **f32)
            _lib.call("paths_linear_f32", p(state_prev), state_prev.stride(1), p(lvl_pack["wh1"]), p(lvl_pack["bh1"]), p(hid), Hh,
                      M, Hh, Hh, D, 1, st)
            _lib.call("paths_linear_f32", p(hid), Hh, p(lvl_pack["wh2"]), p(lvl_pack["bh2"]), p(hctx), D, M, D, D, Hh, 0, st)
        state_out = torch.empty((B, N, D), **f32)
        _lib.call("paths_scale_add_rows", p(fts), p(importance), p(hctx) if hctx is not None else None, p(num_ims), N, D, M,
                  1 if mc.importance_mode == "mul" else 0, p(state_out), st)
        scratch_imp = torch.empty((B, N), **f32)
        importance_proj(state_out, 0, scratch_imp)               # pass 2: tokens = proj_in(Z) + PE

    out = {"ctx_patch": state_out, "importance": importance, "tokens": tokens, "num_ims": num_ims}
    for key in ("keep_idx", "keep_count", "kept_rows"):
        if key in fused:
            out[key] = fused[key]
    if "qkv_img" in fused:
        # the attention's operand images are ready (or, FUSE_QKV = 2, one call away: "qkv_finish" runs on the aggregator's stream)
        out["qkv_img"], out["qkv_finish"], out["_qkv_ws"] = fused["qkv_img"], fused.get("finish"), fused["ws"]
    return out


def parent_partials(lstm_pack, state_out: torch.Tensor, keep_idx: torch.Tensor, keep_count: torch.Tensor,
                    kept_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HP[b*cap + i] = h1[b, keep_idx[b,i]] @ W_gates[:, D:2D]^T  (no bias) for the kept parents of every slide:
    [B*cap, 3Hc+D] in the packed gate-column order.  ~4x fewer rows than the children that will consume it."""
    B, N, Dp = state_out.shape
    Hc = lstm_pack["Hc"]
    D = Dp - Hc
    G = 3 * Hc + D
    cap = keep_idx.shape[1]
    f32 = dict(device=state_out.device, dtype=torch.float32)
    st = _lib.stream()
    p = _lib.ptr
    hp = torch.empty((B * cap, G), **f32)
    if kept_rows is not None:
        # default split mode: the GEMM reads the kept parents' h rows where they are (addresses from paths_topk_rows)
        assert use_x6(D, Hc) and G % 256 == 0 and split_planes() == 2 and kept_rows.shape == (B, cap) and kept_rows.dtype == torch.int64
        wg, wg_s = _x6_of(lstm_pack, "w_gates")
        timed("parent_gemm", lambda: _lib.call("paths_gemm_rows_nt_x6", p(kept_rows), p(wg), 2 * D, D, p(hp), G, B * cap, G, D, 2, wg_s, a_scale(), st))
        return hp
    hk = torch.empty((B * cap, D), **f32)
    _lib.call("paths_gather_kept_rows", p(state_out), N, Dp, p(keep_idx), cap, p(keep_count), D, B, p(hk), st)
    if use_x6(D, Hc) and G % 256 == 0:
        wg, wg_s = _x6_of(lstm_pack, "w_gates")
        _lib.call("paths_gemm_nt_x6", p(hk), D, p(wg), 2 * D, D, None, p(hp), G, B * cap, G, G, D, 0,
                  None, 0, None, 0, 0, split_planes(), wg_s, a_scale(), st)
    else:
        _lib.call("paths_gemm_nt_f32", p(hk), D, lstm_pack["w_gates"].data_ptr() + 4 * D, 2 * D, None, p(hp), G, B * cap, G, G, D, 0,
                  None, 0, None, 0, 0, st)
    return hp


def aggregator_forward(mc, lvl_pack, tokens, num_ims, ctx_prev, ctx_all, status=None, qkv=None) -> Dict[str, torch.Tensor]:
    """The transformer aggregator + classifier of a level (reference model/aggregator.py:58-76, model/paths.py:126-139).
    Nothing here feeds the next level's patch selection, so the device recursion runs it on a second HIP stream.
    ``status`` (optional int32 [1] device tensor): bit 4 is set if a bounded in-launch hand-off wait gave up (csrc/token0_ws.hip).
    ``qkv`` (optional): the dict :func:`selection_forward` returned - when it carries "qkv_img" the first decoder layer's q | k | v
    operand images were written by the importance / projection finish (FUSE_QKV) and the aggregator starts at the attention."""
    qkv_img, qkv_finish = (qkv.get("qkv_img"), qkv.get("qkv_finish")) if qkv is not None else (None, None)
    if qkv_finish is not None:
        timed("agg_tokens_qkv", qkv_finish)            # FUSE_QKV = 2: tokens + images on THIS stream, outside the attention + FFN span
    return timed("aggregator", lambda: _aggregator_forward(mc, lvl_pack, tokens, num_ims, ctx_prev, ctx_all, status, qkv_img),
                 {"T": tokens.shape[1], "d": tokens.shape[2], "L": mc.trans_layers, "planes": split_planes() if GEMM_MODE != "f32" else 0},
                 detail=False)


def _aggregator_forward_ws(mc, lvl_pack, tokens, num_ims, res, cat, depth, qkv_img, q, k, v, xb, ctx_out, logits, token_layer_old, status=None,
                           qkv_ready: bool = False):
    """Default-mode aggregator on the weight-stationary token-layer kernel (csrc/tlayer_ws.hip): in_proj writes the attention
    operand images, attention writes its output as the out_proj operand image, the chain kernel keeps weights in registers and
    shares only activations through LDS."""
    B, T, d = tokens.shape
    H, L = mc.trans_heads, mc.trans_layers
    hd = d // H
    st = _lib.stream()
    p = _lib.ptr
    layers = lvl_pack["layers"]
    qscale = LOG2E / math.sqrt(hd)
    Tp = (T + 63) // 64 * 64
    o_img = torch.empty((B * Tp * d * 4,), device=tokens.device, dtype=torch.uint8)
    g = lambda dct, key: p(dct[key]) if dct is not None else None

    def token_layer(x_in, x_out, post, nxt, use_img=True):
        w = post or nxt
        ip, sp = tlayer_ws_images(post, 0) if post is not None else (None, (1.0, 1.0, 1.0))
        iq, sq = tlayer_ws_images(nxt, 1) if nxt is not None else (None, (1.0,))
        _lib.call("paths_token_layer_ws", p(x_in), None, p(o_img) if post is not None else None, p(x_out) if post is not None else None,
                  p(ip), p(iq), g(post, "bo"), g(post, "ln1g"), g(post, "ln1b"), g(post, "cab"), g(post, "ln2g"), g(post, "ln2b"),
                  g(post, "b1"), g(post, "b2"), g(post, "ln3g"), g(post, "ln3b"), g(nxt, "bqkv"),
                  sp[0], sp[1], sp[2], sq[0], p(qkv_img) if nxt is not None else None, p(num_ims), B, T, d, H,
                  1 if post is not None else 0, 1 if nxt is not None else 0, 1, qscale, w["eps"], None, 0, st)

    xa = tokens
    if not qkv_ready:                       # (FUSE_QKV: the importance / projection finish already wrote layer 0's operand images)
        timed("agg_in_proj", lambda: token_layer(xa, None, None, layers[0]))
    for l in range(L - 1):
        timed("agg_attention", lambda: _lib.call("paths_attention_h3_img", p(o_img), p(num_ims), B, T, H, hd, p(qkv_img), st))
        last = l + 1 == L - 1
        timed("agg_token_chain", lambda: token_layer(xa, xb, layers[l], None if last else layers[l + 1]))
        xa, xb = xb, xa
    w = layers[L - 1]

    def tail_ws():
        img = token0_ws_image(w, qscale)
        part = torch.empty((int(_lib.load().paths_token0_ws_partials_d(B, T, d)),), device=tokens.device, dtype=torch.float32)
        cnt = token0_counters(tokens.device, B)
        _lib.call(
            "paths_token0_tail_ws", p(xa), p(num_ims), p(img), w["bqkv"].data_ptr() + 4 * 2 * d, p(w["bo"]), p(w["ln1g"]), p(w["ln1b"]),
            p(w["cab"]), p(w["ln2g"]), p(w["ln2b"]), p(w["b1"]), p(w["b2"]), p(w["ln3g"]), p(w["ln3b"]),
            p(lvl_pack["lnfg"]), p(lvl_pack["lnfb"]), p(res) if res is not None else None,
            res.stride(0) if res is not None else 0, p(cat) if cat is not None else None, depth,
            p(lvl_pack["wcls"]), p(lvl_pack["bcls"]), logits.shape[1], lvl_pack["wcls"].shape[1], p(ctx_out), p(logits),
            p(part), p(cnt), p(status) if status is not None else None, B, T, d, H, w["eps"], lvl_pack["lnf_eps"], 1 if qkv_ready else 0, st)

    if TAIL_WS:
        timed("agg_token0_tail", tail_ws)
        return {"logits": logits, "ctx_slide": ctx_out}

    def tail():
        token_layer_old(xa, None, None, w)          # fp32 q, k, v of the last layer for the token-0 tail
        ws_part = torch.empty((B * H * 16 * 36,), device=tokens.device, dtype=torch.float32)
        _lib.call(
            "paths_token0_tail", p(xa), p(q), p(k), p(v), p(num_ims), p(w["wo"]), p(w["bo"]), p(w["ln1g"]), p(w["ln1b"]),
            p(w["cab"]), p(w["ln2g"]), p(w["ln2b"]), p(w["w1"]), p(w["b1"]), p(w["w2"]), p(w["b2"]), p(w["ln3g"]), p(w["ln3b"]),
            p(lvl_pack["lnfg"]), p(lvl_pack["lnfb"]), p(res) if res is not None else None,
            res.stride(0) if res is not None else 0, p(cat) if cat is not None else None, depth,
            p(lvl_pack["wcls"]), p(lvl_pack["bcls"]), logits.shape[1], lvl_pack["wcls"].shape[1], p(ctx_out), p(logits),
            p(ws_part), B, T, d, H, w["eps"], lvl_pack["lnf_eps"], st)

    timed("agg_token0_tail", tail)
    return {"logits": logits, "ctx_slide": ctx_out}


def _aggregator_forward(mc, lvl_pack, tokens, num_ims, ctx_prev, ctx_all, status=None, qkv_img=None) -> Dict[str, torch.Tensor]:
    _lib.require_cuda(tokens, num_ims, ctx_prev, ctx_all)
    if AGG_FP8:
        return _aggregator_forward_generic(mc, lvl_pack, tokens, num_ims, ctx_prev, ctx_all, fp8=True)
    if not fast_path(mc):
        return _aggregator_forward_generic(mc, lvl_pack, tokens, num_ims, ctx_prev, ctx_all, status=status)
    B, T, d = tokens.shape
    H, L = mc.trans_heads, mc.trans_layers
    st = _lib.stream()
    p = _lib.ptr
    f32 = dict(device=tokens.device, dtype=torch.float32)
    hd = d // H
    q = torch.empty((B, H, T, hd), **f32)
    k = torch.empty((B, H, T, hd), **f32)
    v = torch.empty((B, H, T, hd), **f32)
    attn = torch.empty((B, T, d), **f32)
    xa, xb = tokens, torch.empty((B, T, d), **f32)
    qscale = LOG2E / math.sqrt(hd)
    layers = lvl_pack["layers"]

    def token_layer(x_in, x_out, post, nxt, max_tokens=0, qkv_images=None):
        w = post or nxt
        g = lambda dct, key: p(dct[key]) if dct is not None else None
        if GEMM_MODE == "h3":
            ip, sp = tlayer_h3_images(post, 0) if post is not None else (None, (1.0, 1.0, 1.0))
            iq, sq = tlayer_h3_images(nxt, 1) if nxt is not None else (None, (1.0,))
            _lib.call("paths_token_layer_h3", p(x_in), p(attn) if post else None, p(x_out) if post else None, p(ip), p(iq),
                      g(post, "bo"), g(post, "ln1g"), g(post, "ln1b"), g(post, "cab"), g(post, "ln2g"), g(post, "ln2b"),
                      g(post, "b1"), g(post, "b2"), g(post, "ln3g"), g(post, "ln3b"), g(nxt, "bqkv"),
                      sp[0], sp[1], sp[2], sq[0], p(q), p(k), p(v), p(num_ims), B, T, d, H,
                      1 if post else 0, 1 if nxt else 0, 1, qscale, w["eps"], max_tokens, p(qkv_images), st)
            return
        _lib.call("paths_token_layer_f32", p(x_in), p(attn) if post else None, p(x_out) if post else None,
                  g(post, "wo"), g(post, "bo"), g(post, "ln1g"), g(post, "ln1b"), g(post, "cab"), g(post, "ln2g"), g(post, "ln2b"),
                  g(post, "w1"), g(post, "b1"), g(post, "w2"), g(post, "b2"), g(post, "ln3g"), g(post, "ln3b"),
                  g(nxt, "wqkv"), g(nxt, "bqkv"), p(q), p(k), p(v), p(num_ims), B, T, d, H,
                  1 if post else 0, 1 if nxt else 0, 1, qscale, w["eps"], max_tokens, st)

    nlog = lvl_pack["wcls"].shape[0]
    ctx_out = torch.empty((B, d), **f32)
    logits = torch.empty((B, nlog), **f32)
    res = ctx_prev if mc.slide_ctx_mode == "residual" else None
    cat = ctx_all.contiguous() if (mc.slide_ctx_mode == "concat" and ctx_all is not None and ctx_all.shape[1] > 0) else None
    depth = cat.shape[1] if cat is not None else 0

    attn_ws = None
    if qkv_img is not None:
        assert GEMM_MODE == "h3" and TLAYER_WS and QKV_IMAGES and L > 1 and not ATTN_FP8
        return _aggregator_forward_ws(mc, lvl_pack, tokens, num_ims, res, cat, depth, qkv_img, q, k, v, xb, ctx_out, logits, token_layer, status,
                                      qkv_ready=True)
    fp8 = ATTN_FP8 and L > 1          # opt-in e4m3 attention (csrc/attn_fp8.hip: outside the 1e-4 logit bar, stress-config measurement only)
    if fp8:
        attn_ws = torch.empty((int(_lib.load().paths_attention_fp8_workspace(B, T, H, hd)),), device=tokens.device, dtype=torch.uint8)
    elif GEMM_MODE != "f32" and L > 1:
        attn_ws = torch.empty((int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, split_planes())),),
                              device=tokens.device, dtype=torch.uint8)
    # default mode: the in_proj of a layer that feeds the full attention writes the attention kernel's operand images itself
    # (no fp32 q, k, v round trip, no re-write launch); the last layer's q, k, v stay fp32 for the token-0 tail
    direct = GEMM_MODE == "h3" and attn_ws is not None and QKV_IMAGES and not fp8
    if direct and TLAYER_WS:
        return _aggregator_forward_ws(mc, lvl_pack, tokens, num_ims, res, cat, depth, attn_ws, q, k, v, xb, ctx_out, logits, token_layer, status)
    timed("agg_in_proj", lambda: token_layer(xa, None, None, layers[0], qkv_images=attn_ws if direct else None))
    for l in range(L - 1):
        if fp8:
            timed("agg_attention", lambda: _lib.call("paths_attention_fp8", p(q), p(k), p(v), p(attn), p(num_ims), B, T, H, hd, p(attn_ws), st))
        elif GEMM_MODE != "f32":
            timed("agg_attention", lambda: _lib.call("paths_attention_x6", p(q), p(k), p(v), p(attn), None, p(num_ims), B, T, H, hd, 0,
                                                     p(attn_ws), split_planes(), 1 if direct else 0, st))
        else:
            timed("agg_attention", lambda: _lib.call("paths_attention_f32", p(q), p(k), p(v), p(attn), None, p(num_ims), B, T, H, hd, 0, st))
        timed("agg_token_chain", lambda: token_layer(xa, xb, layers[l], layers[l + 1], qkv_images=attn_ws if (direct and l + 1 < L - 1) else None))
        xa, xb = xb, xa
    # Last layer: only token 0 of its output is read (aggregator.py:75) -> one fused launch per level computes the
    # single-query attention, the row chain, decoder.norm, the slide-context residual and the classifier.
    w = layers[L - 1]
    ws_part = torch.empty((B * H * 16 * 36,), **f32)
    timed("agg_token0_tail", lambda: _lib.call(
        "paths_token0_tail", p(xa), p(q), p(k), p(v), p(num_ims), p(w["wo"]), p(w["bo"]), p(w["ln1g"]), p(w["ln1b"]),
        p(w["cab"]), p(w["ln2g"]), p(w["ln2b"]), p(w["w1"]), p(w["b1"]), p(w["w2"]), p(w["b2"]), p(w["ln3g"]), p(w["ln3b"]),
        p(lvl_pack["lnfg"]), p(lvl_pack["lnfb"]), p(res) if res is not None else None,
        res.stride(0) if res is not None else 0, p(cat) if cat is not None else None, depth,
        p(lvl_pack["wcls"]), p(lvl_pack["bcls"]), nlog, lvl_pack["wcls"].shape[1], p(ctx_out), p(logits),
        p(ws_part), B, T, d, H, w["eps"], lvl_pack["lnf_eps"], st))
    return {"logits": logits, "ctx_slide": ctx_out}
