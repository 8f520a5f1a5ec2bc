"""Counter-based synthetic slide bags and model weights (SURVEY.md §8d "Synthetic inputs").

Everything here is a pure function of integer counters, so the same bytes can be produced
 * on the host with numpy (this file) and
 * on the device by ``paths_synth_grid`` (paths_amd/csrc/synth.hip),
without storing multi-GB feature grids.  No transcendental is involved: the generator is
bit-identical on CPU and GPU.

Definition (all arithmetic mod 2**32)::

    fmix32(h): h ^= h>>16; h *= 0x85EBCA6B; h ^= h>>13; h *= 0xC2B2AE35; h ^= h>>16
    k0   = fmix32(seed ^ 0x5BD1E995)
    k1   = fmix32(k0 + slide * 0x9E3779B1)
    k2   = fmix32(k1 + level * 0x85EBCA77 + 1)
    k3   = fmix32(k2 + x * 0xC2B2AE3D)
    cell = fmix32(k3 + y * 0x27D4EB2F)
    u    = fmix32(cell + c * 0x165667B1 + 0x9E3779B9)
    feature[x, y, c] = float32(((u >> 8) * 2**-23 - 1)) * float32(sqrt(3))     # uniform[-sqrt3, sqrt3)
    background(x, y) = level >= 1 and fmix32(cell ^ 0xB6B6B6B6) < p_bg * 2**32    # all-zero row

The grid format is the reference's preprocessed-grid format: ``[X, Y, D]`` float32 per level, an
all-zero row marks a background cell (reference preprocess/preprocess.py:89,172-175); level ``l``
has ``2**l`` times the level-0 cells per side (reference data_utils/slide.py:303-315).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)
SQRT3_F32 = np.float32(math.sqrt(3.0))


def fmix32(h: np.ndarray) -> np.ndarray:
    """murmur3 finaliser on uint64 arrays holding 32-bit values."""
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def _u64(a) -> np.ndarray:
    return np.asarray(a, dtype=np.uint64)


def slide_level_key(seed: int, slide: int, level: int) -> np.uint64:
    k0 = fmix32(_u64(seed & 0xFFFFFFFF) ^ np.uint64(0x5BD1E995))
    k1 = fmix32(k0 + _u64(slide) * np.uint64(0x9E3779B1))
    k2 = fmix32(k1 + _u64(level) * np.uint64(0x85EBCA77) + np.uint64(1))
    return np.uint64(k2)


def cell_key(seed: int, slide: int, level: int, x, y) -> np.ndarray:
    k2 = slide_level_key(seed, slide, level)
    k3 = fmix32(k2 + _u64(x) * np.uint64(0xC2B2AE3D))
    return fmix32(k3 + _u64(y) * np.uint64(0x27D4EB2F))


def bg_threshold(p_bg: float) -> int:
    """Integer threshold t such that a cell is background iff hash < t."""
    return int(round(p_bg * 4294967296.0))


def cell_is_background(seed: int, slide: int, level: int, x, y, p_bg: float) -> np.ndarray:
    x = np.asarray(x)
    if level == 0 or p_bg <= 0.0:
        return np.zeros(x.shape, dtype=bool)
    ck = cell_key(seed, slide, level, x, y)
    return fmix32(ck ^ np.uint64(0xB6B6B6B6)) < np.uint64(bg_threshold(p_bg))


def u32_to_feature(u: np.ndarray) -> np.ndarray:
    v = ((u >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -23)) - np.float32(1.0)
    return (v * SQRT3_F32).astype(np.float32)


def cell_features(seed: int, slide: int, level: int, x, y, dim: int, p_bg: float) -> np.ndarray:
    """Feature rows [n, dim] float32 for cells (x[i], y[i]); background cells are all-zero."""
    x = np.asarray(x, dtype=np.int64).reshape(-1)
    y = np.asarray(y, dtype=np.int64).reshape(-1)
    ck = cell_key(seed, slide, level, x, y)                       # [n]
    c = np.arange(dim, dtype=np.uint64) * np.uint64(0x165667B1) + np.uint64(0x9E3779B9)
    u = fmix32(ck[:, None] + c[None, :])
    f = u32_to_feature(u)
    bg = cell_is_background(seed, slide, level, x, y, p_bg)
    f[bg] = 0.0
    return f


@dataclass
class SyntheticSlide:
    """A lazily evaluated 5-level feature pyramid; ``grid(level)`` materialises a dense [X,Y,D]."""
    seed: int
    slide: int
    base_shape: Tuple[int, int]
    dim: int = 1024
    num_levels: int = 5
    p_bg: float = 0.1

    def shape(self, level: int) -> Tuple[int, int]:
        return self.base_shape[0] << level, self.base_shape[1] << level

    def rows(self, level: int, x, y) -> np.ndarray:
        return cell_features(self.seed, self.slide, level, x, y, self.dim, self.p_bg)

    def is_background(self, level: int, x, y) -> np.ndarray:
        return cell_is_background(self.seed, self.slide, level, x, y, self.p_bg)

    def grid(self, level: int) -> np.ndarray:
        X, Y = self.shape(level)
        xs, ys = np.meshgrid(np.arange(X), np.arange(Y), indexing="ij")
        out = np.empty((X * Y, self.dim), dtype=np.float32)
        step = max(1, (1 << 22) // self.dim)
        xs, ys = xs.reshape(-1), ys.reshape(-1)
        for i in range(0, X * Y, step):
            out[i:i + step] = self.rows(level, xs[i:i + step], ys[i:i + step])
        return out.reshape(X, Y, self.dim)

    def label(self, nbins: int = 4) -> Tuple[int, int]:
        """(survival_bin, censored) — SURVEY.md §8d: hash % 4, hash % 2."""
        h = int(fmix32(slide_level_key(self.seed, self.slide, 0xFFFF) ^ np.uint64(0x1ABE1)))
        return h % nbins, (h >> 8) % 2


# ---------------------------------------------------------------------------------------------
# Weights: one uniform stream per tensor name, bound chosen like torch's default initialisers.
# ---------------------------------------------------------------------------------------------

def _name_key(seed: int, name: str) -> np.uint64:
    h = np.uint64(fmix32(_u64(seed & 0xFFFFFFFF) ^ np.uint64(0x7F4A7C15)))
    for ch in name.encode():
        h = np.uint64(fmix32(h * np.uint64(31) + np.uint64(ch)))
    return h


def uniform_tensor(seed: int, name: str, shape, bound: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    key = _name_key(seed, name)
    idx = np.arange(n, dtype=np.uint64)
    u = fmix32(fmix32(key + idx * np.uint64(0x9E3779B1)) + np.uint64(0x85EBCA77))
    v = ((u >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -23)) - np.float32(1.0)
    return (v * np.float32(bound)).astype(np.float32).reshape(shape)


def make_state_dict(seed: int, shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, np.ndarray]:
    """Generate a full reference-keyed state_dict from tensor shapes.

    Linear weights/biases: uniform(±1/sqrt(fan_in)); transformer in_proj/linear matrices:
    xavier-uniform bound; LayerNorm weight = 1, bias = 0; special_token uniform·sqrt3;
    ``multihead_attn.out_proj.bias`` deliberately NON-zero (it is the only live part of the
    degenerate cross-attention).
    """
    out = {}
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if ".norm" in name:
            out[name] = (np.ones if leaf == "weight" else np.zeros)(shape, dtype=np.float32)
            continue
        if name.endswith("special_token"):
            out[name] = uniform_tensor(seed, name, shape, math.sqrt(3.0))
            continue
        if leaf in ("weight", "in_proj_weight"):
            fan_out, fan_in = shape
            if ".transformer." in name:
                bound = math.sqrt(6.0 / (fan_in + fan_out))
            else:
                bound = 1.0 / math.sqrt(fan_in)
            out[name] = uniform_tensor(seed, name, shape, bound)
        else:  # biases
            wname = name[: -len(leaf)] + ("in_proj_weight" if leaf == "in_proj_bias" else "weight")
            fan_in = shapes[wname][1] if wname in shapes else shape[0]
            out[name] = uniform_tensor(seed, name, shape, 1.0 / math.sqrt(fan_in))
    return out
