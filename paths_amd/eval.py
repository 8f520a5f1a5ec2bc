"""Evaluators of the train/eval harness (reference eval.py:9-120) with self-contained metrics.

The reference delegates to ``sksurv.metrics.concordance_index_censored`` and ``torcheval.metrics.BinaryAUROC``; neither
is a dependency here, so both metrics are restated from their published definitions:

* c-index (Harrell, as implemented by scikit-survival): a pair (i, j) is comparable when sample i had an EVENT and
  time_i < time_j, or time_i == time_j and j is censored; it is concordant when risk_i > risk_j, counts 1/2 when
  |risk_i - risk_j| <= 1e-8; c = (concordant + tied/2) / comparable.
* AUROC: area under the ROC curve with tied scores on a diagonal segment = P(s+ > s-) + P(s+ = s-)/2; 0.5 when a class
  is absent (torcheval's convention).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, Optional

import numpy as np
import torch


def concordance_index_censored(event: np.ndarray, time: np.ndarray, risk: np.ndarray, tied_tol: float = 1e-8) -> float:
    event = np.asarray(event, dtype=bool)
    time = np.asarray(time, dtype=np.float64)
    risk = np.asarray(risk, dtype=np.float64)
    ti, tj = time[:, None], time[None, :]
    comparable = event[:, None] & ((ti < tj) | ((ti == tj) & ~event[None, :]))
    np.fill_diagonal(comparable, False)
    n = comparable.sum()
    if n == 0:
        raise ValueError("no comparable pairs")
    diff = risk[:, None] - risk[None, :]
    tied = comparable & (np.abs(diff) <= tied_tol)
    conc = comparable & (diff > tied_tol)
    return float((conc.sum() + 0.5 * tied.sum()) / n)


def binary_auroc(scores: np.ndarray, target: np.ndarray) -> float:
    scores = np.asarray(scores, dtype=np.float64)
    target = np.asarray(target).astype(bool)
    npos, nneg = int(target.sum()), int((~target).sum())
    if npos == 0 or nneg == 0:
        return 0.5
    order = np.argsort(scores, kind="stable")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):                      # average ranks over ties
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty(len(s), dtype=np.float64)
    r[order] = ranks
    return float((r[target].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def _gather_np(a: np.ndarray) -> np.ndarray:
    import torch.distributed as dist
    if not dist.is_initialized():
        return a
    objs = [None] * dist.get_world_size()
    dist.all_gather_object(objs, a)
    return np.concatenate(objs)


def _cat(chunks, tail=()) -> np.ndarray:
    """np.concatenate that also works on a rank that registered nothing (it still has to take part in the gather)."""
    return np.concatenate(chunks) if chunks else np.zeros((0,) + tuple(tail), np.float64)


class Evaluator(ABC):
    def __init__(self, split: str):
        self.losses, self.weights = [], []
        self.split = split

    @abstractmethod
    def reset(self):
        ...

    @abstractmethod
    def register(self, batch, outputs, loss):
        ...

    @abstractmethod
    def calculate(self, train_stats=None, epoch=None) -> Dict:
        ...

    def _mean_loss(self) -> float:
        l, w = _gather_np(np.asarray(self.losses, np.float64)), _gather_np(np.asarray(self.weights, np.float64))
        return float((l * w).sum() / max(w.sum(), 1e-30))

    def _add_to_train_stats(self, epoch, out, train_stats):          # reference eval.py:27-35
        if train_stats is not None:
            for key in out:
                if key in train_stats:
                    if epoch is None:
                        train_stats[key] = out[key]
                    else:
                        train_stats[key][epoch] = out[key]


class SurvivalEvaluator(Evaluator):
    """reference eval.py:38-84: risk = -sum_t cumprod(1 - hazards); c-index over the registered samples."""

    def __init__(self, split: str):
        super().__init__(split)
        self.cens, self.times, self.risks = [], [], []

    def reset(self):
        for l in (self.losses, self.weights, self.cens, self.times, self.risks):
            l.clear()

    def register(self, batch, hazards, loss, weight: Optional[float] = None):
        self.losses.append(float(loss))
        self.weights.append(float(weight if weight is not None else hazards.shape[0]))
        survival = torch.cumprod(1 - hazards.detach(), dim=1)
        self.risks.append((-survival.sum(dim=1)).cpu().numpy())
        self.cens.append(torch.as_tensor(batch["censored"]).cpu().numpy())
        self.times.append(torch.as_tensor(batch["survival"]).cpu().numpy())

    def calculate(self, train_stats=None, epoch=None):
        event = (1 - _gather_np(_cat(self.cens))).astype(bool)
        times, risks = _gather_np(_cat(self.times)), _gather_np(_cat(self.risks))
        if event.sum() <= 1:
            c_index = 0.5
        else:
            c_index = concordance_index_censored(event, times, risks)
        out = {f"{self.split}_loss": self._mean_loss(), f"{self.split}_c-index": c_index}
        self._add_to_train_stats(epoch, out, train_stats)
        return out


class SubtypeClassificationEvaluator(Evaluator):
    """reference eval.py:87-120: mean one-vs-rest AUROC of the softmax scores."""

    def __init__(self, split: str, nclasses: int):
        super().__init__(split)
        self.nclasses = nclasses
        self.preds, self.labels = [], []

    def reset(self):
        for l in (self.losses, self.weights, self.preds, self.labels):
            l.clear()

    def register(self, batch, logits, loss, weight: Optional[float] = None):
        self.losses.append(float(loss))
        self.weights.append(float(weight if weight is not None else logits.shape[0]))
        self.preds.append(torch.softmax(logits.detach(), dim=-1).cpu().numpy())
        self.labels.append(torch.as_tensor(batch["subtype"]).cpu().numpy())

    def calculate(self, train_stats=None, epoch=None):
        preds, labels = _gather_np(_cat(self.preds, (self.nclasses,))), _gather_np(_cat(self.labels))
        aucs = [binary_auroc(preds[:, i], labels == i) for i in range(self.nclasses)]
        out = {f"{self.split}_loss": self._mean_loss(), f"{self.split}_AUC": float(sum(aucs) / len(aucs))}
        self._add_to_train_stats(epoch, out, train_stats)
        return out
