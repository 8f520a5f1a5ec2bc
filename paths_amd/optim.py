"""AdamW on the device in ONE launch per step (csrc/optim.hip), as a drop-in subclass of ``torch.optim.AdamW``.

Reference: train.py:49-50 builds ``torch.optim.AdamW(model.parameters(), lr, weight_decay)`` and train.py:66 calls ``opt.step()``.
torch's default (foreach) implementation costs 3-4 ms of HOST time per step over the ~300 parameter tensors of a 5-level model -
a fifth of a training step whose host and device sides are equally long.  ``HipAdamW.step()`` walks the parameters once, writes a
pointer table and calls ``paths_adamw_multi``; the kernel follows the foreach implementation's operation order and rounding, so
trajectories are the ones ``torch.optim.AdamW`` produces bit for bit (test_hip_adamw_is_bitwise_torch_foreach; fixture G10).
State layout, ``state_dict()`` and ``load_state_dict()`` are torch's own ('step', 'exp_avg', 'exp_avg_sq' per parameter).
Anything the kernel does not cover (amsgrad, maximize, capturable, non-fp32 / non-contiguous tensors, 1 - beta1 >= 0.5) takes
torch's implementation unchanged."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import _lib

FLAVOR = 7     # lerp / addcmul / addcdiv as fused multiply-adds: what the installed torch build's kernels compute (see the test)


class HipAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._maps: Dict[tuple, dict] = {}          # per (group, set of stepped parameters): block map + device staging
        self._last_key = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._maps.clear()                          # (the maps cache the state tensors' addresses and step counters)
        self._last_key = None

    def _eligible(self, group) -> bool:
        lr = group["lr"]
        return (not group.get("amsgrad", False) and not group.get("maximize", False) and not group.get("capturable", False)
                and not group.get("differentiable", False) and not isinstance(lr, torch.Tensor)
                and 0.0 < 1.0 - group["betas"][0] < 0.5)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            ok = self._eligible(group) and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                                               and p.grad.is_contiguous() and not p.grad.is_sparse for p in ps)
            if not ok:
                self._torch_step(group)
                continue
            self._hip_step(gi, group, ps)
        return loss

    def _torch_step(self, group):
        """torch's own implementation for one group (whatever the kernel does not cover)."""
        saved = self.param_groups
        try:
            self.param_groups = [group]
            torch.optim.AdamW.step(self)
        finally:
            self.param_groups = saved

    def _flatten_steps(self, ps: List[torch.nn.Parameter]) -> torch.Tensor:
        """The 'step' counters of ``ps`` as 0-dim VIEWS of one flat fp32 CPU tensor (values kept): state_dict() / torch's own
        implementation see ordinary scalar tensors, the per-step increment is one add."""
        state = self.state
        flat = torch.tensor([float(state[p]["step"]) for p in ps], dtype=torch.float32)
        for i, p in enumerate(ps):
            state[p]["step"] = flat[i]
        return flat

    def _hip_step(self, gi: int, group, ps: List[torch.nn.Parameter]):
        dev = ps[0].device
        state = self.state
        new = [p for p in ps if len(state[p]) == 0]
        if new:
            # exp_avg / exp_avg_sq of the newcomers as views of two flat zero buffers (one fill each instead of two per parameter);
            # step counters are CPU scalars as in torch's non-capturable default
            tot = sum(p.numel() for p in new)
            flat_m, flat_v = torch.zeros((tot,), device=dev, dtype=torch.float32), torch.zeros((tot,), device=dev, dtype=torch.float32)
            off = 0
            for p in new:
                n = p.numel()
                st = state[p]
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = flat_m[off:off + n].view_as(p)
                st["exp_avg_sq"] = flat_v[off:off + n].view_as(p)
                off += n
        key = (gi, tuple(id(p) for p in ps))
        mp = self._maps.get(key)
        if mp is None:
            chunk = int(_lib.load().paths_adamw_chunk())
            blocks = []
            for t, p in enumerate(ps):
                blocks += [(t, e0) for e0 in range(0, p.numel(), chunk)]
            n = len(ps)
            mp = {"n": n, "nblocks": len(blocks),
                  "blocks": torch.tensor(blocks, dtype=torch.int32, device=dev).contiguous(),
                  "numel": torch.tensor([p.numel() for p in ps], dtype=torch.int64, device=dev),
                  # host staging: pinned, 4 rotating slots PER MAP, each with the event recorded behind its last host-to-device copy;
                  # the host waits for that event before it rewrites the slot (a loop that never syncs, or several parameter
                  # groups stepping through one optimizer, must not let a queued copy pick up a later step's table)
                  "steps_flat": self._flatten_steps(ps), "step_vals": np.array([float(state[p]["step"]) for p in ps], dtype=np.float64),
                  "m_ptrs": [state[p]["exp_avg"].data_ptr() for p in ps], "v_ptrs": [state[p]["exp_avg_sq"].data_ptr() for p in ps],
                  "host": [torch.empty((6 * n,), dtype=torch.int64).pin_memory() for _ in range(4)],
                  "dev": [torch.empty((6 * n,), dtype=torch.int64, device=dev) for _ in range(4)],
                  "copied": [None] * 4, "slot": 0}
            self._maps[key] = mp
        n = mp["n"]
        if self._last_key != key:                    # another parameter set stepped in between: its counters moved (and it may have
            mp["steps_flat"] = self._flatten_steps(ps)   # re-pointed some of them into ITS flat tensor): gather them again
            mp["step_vals"] = mp["steps_flat"].numpy().astype(np.float64)
            self._last_key = key
        mp["steps_flat"].add_(1.0)                   # the state's own step counters: CPU scalars as torch keeps them, here as views of
        mp["step_vals"] += 1.0                       # ONE flat tensor (a foreach add over ~170 scalar tensors was 0.6 ms of host time per step)
        lr, (beta1, beta2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
        # scalars exactly as torch/optim/adam.py computes them (Python floats = doubles, rounded to fp32 when they enter a kernel)
        svals = mp["step_vals"]
        slot = mp["slot"] = (mp["slot"] + 1) & 3
        host, devbuf = mp["host"][slot], mp["dev"][slot]
        if mp["copied"][slot] is not None:
            mp["copied"][slot].synchronize()         # (four steps old: a no-op unless the device is that far behind)
        h = host.numpy()
        tab = h[:4 * n].reshape(n, 4)
        tab[:, 0] = [p.data_ptr() for p in ps]
        tab[:, 1] = [p.grad.data_ptr() for p in ps]
        tab[:, 2] = mp["m_ptrs"]
        tab[:, 3] = mp["v_ptrs"]
        f = h[4 * n:6 * n].view(np.float32)          # 2 n int64 words hold 4 n floats: [0, n) step_size, [n, 2n) bc2_sqrt
        if svals[0] == svals[-1] and (svals == svals[0]).all():
            t = float(svals[0])
            f[:n] = (lr / (1 - beta1 ** t)) * -1
            f[n:2 * n] = (1 - beta2 ** t) ** 0.5
        else:                                        # parameters that skipped steps (grad None now and then) lag behind
            f[:n] = [(lr / (1 - beta1 ** float(t))) * -1 for t in svals]
            f[n:2 * n] = [(1 - beta2 ** float(t)) ** 0.5 for t in svals]
        devbuf.copy_(host, non_blocking=True)
        if mp["copied"][slot] is None:
            mp["copied"][slot] = torch.cuda.Event()
        mp["copied"][slot].record()
        base = devbuf.data_ptr()
        _lib.call("paths_adamw_multi", base, _lib.ptr(mp["numel"]), _lib.ptr(mp["blocks"]), mp["nblocks"], base + 32 * n, base + 32 * n + 4 * n,
                  1 - lr * wd, 1 if wd != 0 else 0, 1 - beta1, beta2, 1 - beta2, eps, FLAVOR, _lib.stream())
        # The kernel writes the parameters through raw addresses: tell torch (and everything keyed on the version counters - the
        # cached weight images of paths_amd/ops.py, recorded launch tapes, captured graphs) that they changed, as an in-place torch
        # op would.  (Round 3 saw the reference's epoch loop, fixture G10, diverge under torch's FUSED AdamW and blamed rounding;
        # the same 0.570-vs-0.539 epoch-2 loss appeared here until the counters were bumped: stale weight images, not arithmetic.)
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
