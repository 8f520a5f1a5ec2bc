"""Train / eval harness around the device-resident recursion (reference train.py:31-116, utils.py:169-198).

Same semantics as the reference loop: AdamW(lr, weight_decay) + ExponentialLR(lr_decay_per_epoch), shuffled batches
of ``config.batch_size[0]`` slides, validation every ``eval_epochs`` with early stopping on c-index / AUC, final test
evaluation, ``model.pt`` (= ``state_dict``) + ``train_stats.pkl`` in the reference's format (checkpoints are
interchangeable), resume from ``train_stats["epoch"]``.  Differences: wandb is replaced by JSON lines, slides are
:class:`DeviceSlide` objects that stay resident in HBM, and with ``WORLD_SIZE > 1`` every global batch is sharded over
the ranks (same permutation on every rank, gradients summed by one RCCL all-reduce).

    python -m paths_amd.train -m MODEL_DIR --synthetic 32     # demo / smoke on synthetic slides
"""
from __future__ import annotations

import json
import os
import pickle
from typing import Dict, List, Optional, Sequence

import torch

from . import distributed as pdist
from . import utils as putils
from .data_utils.slide import DeviceSlide, DeviceSlideBatch
from .eval import SubtypeClassificationEvaluator, SurvivalEvaluator


def save_state(root_path: str, model, train_stats):
    """reference utils.py:169-178"""
    # written under a temporary name and renamed: a reader (another rank re-loading the early-stopping checkpoint) sees the old
    # file or the new one, never a half-written one
    tmp = os.path.join(root_path, "model.pt.tmp")
    torch.save(model.state_dict(), tmp)
    os.replace(tmp, os.path.join(root_path, "model.pt"))
    tmp = os.path.join(root_path, "train_stats.pkl.tmp")
    with open(tmp, "wb") as fh:
        pickle.dump(train_stats, fh)
    os.replace(tmp, os.path.join(root_path, "train_stats.pkl"))


def load_state(root_path: str, model, map_location=None) -> Dict:
    """reference utils.py:181-198"""
    model_path = os.path.join(root_path, "model.pt")
    stats_path = os.path.join(root_path, "train_stats.pkl")
    if os.path.isfile(model_path):
        model.load_state_dict(torch.load(model_path, map_location=map_location))
        from . import ops
        ops.reset_lagged_scales()          # the training forward's fp16 weight scales lag one step behind the weights
    if not os.path.isfile(stats_path):
        return {"epoch": 1}
    with open(stats_path, "rb") as fh:
        return pickle.load(fh)


def epoch_permutation(n: int, shuffle: bool) -> List[int]:
    """Order of one pass over a dataset, consuming the global CPU generator exactly like iterating the reference's
    ``DataLoader`` does (train.py:19-28): EVERY iterator first draws its worker base seed (the validation / test loaders too -
    one draw per evaluation pass, which shifts the next epoch's shuffle; fixture G10 pins this), a shuffling one then draws the
    RandomSampler's seed and runs ``torch.randperm`` with a private generator.  So ``torch.manual_seed(s)`` gives the reference's
    sample order across a whole training run."""
    torch.empty((), dtype=torch.int64).random_()      # the DataLoader iterator draws its worker base seed first
    if not shuffle:
        return list(range(n))
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


def iterate_batches(dataset: Sequence[dict], batch_size: int, shuffle: bool, rank: int, world: int):
    """Yields (local batch dict, global batch size).  Every rank walks the same global batches and keeps its shard."""
    order = epoch_permutation(len(dataset), shuffle)
    for s in range(0, len(order), batch_size):
        glob = order[s:s + batch_size]
        mine = [glob[i] for i in pdist.shard_range(len(glob), rank, world)]
        if not mine:
            yield None, len(glob)
            continue
        items = [dataset[i] for i in mine]
        batch = {"slide": DeviceSlideBatch([it["slide"] for it in items])}
        for key in ("survival_bin", "survival", "censored", "subtype"):
            if key in items[0]:
                batch[key] = torch.as_tensor([it[key] for it in items])
        yield batch, len(glob)


def _evaluate(model, dataset, config, evaluator, rank, world):
    model.eval()
    with torch.no_grad():
        for batch, gb in iterate_batches(dataset, config.batch_size[0], False, rank, world):
            if batch is None:
                continue
            out = putils.recurse(model, batch["slide"], config.top_k_patches, config.num_levels)
            outputs, loss = putils.loss_from_logits(out["logits"], batch, config.task)
            evaluator.register(batch, outputs, loss)


def train_loop(model, train_ds, val_ds, test_ds, config, model_dir: str, log=None) -> Dict:
    """reference train.py:31-116.  Datasets are sequences of dicts {"slide": DeviceSlide, labels...}."""
    rank, world, _ = pdist.env_rank_world()
    log = log or (lambda d: print(json.dumps(d), flush=True) if rank == 0 else None)

    def mk_eval(split):
        if config.task == "subtype_classification":
            return SubtypeClassificationEvaluator(split, len(config.filter_to_subtypes))
        return SurvivalEvaluator(split)

    score_key = "c-index" if config.task == "survival" else "AUC"
    train_stats = load_state(model_dir, model)
    start_epoch = train_stats["epoch"]
    for key in ["train_loss", f"train_{score_key}", "val_loss", f"val_{score_key}"]:
        train_stats.setdefault(key, {})
    train_eval, val_eval = mk_eval("train"), mk_eval("val")
    # reference train.py:49-50: torch.optim.AdamW.  HipAdamW IS that class (state, state_dict, schedulers) with step() as one HIP
    # launch that reproduces torch's foreach update bit for bit (paths_amd/optim.py); PATHS_TORCH_ADAMW=1 keeps torch's step
    if next(model.parameters()).is_cuda and os.environ.get("PATHS_TORCH_ADAMW", "0") == "0":
        from .optim import HipAdamW
        opt = HipAdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay)
    sched = config.get_lr_scheduler(opt)
    ar = pdist.allreduce_gradients if world > 1 else None
    best_val = -1
    model.train()
    for e in range(start_epoch, config.num_epochs + 1):
        for batch, gb in iterate_batches(train_ds, config.batch_size[0], True, rank, world):
            if batch is None:
                # this rank holds no slide of a short last batch: zeros for exactly the gradient set of an active rank (the
                # unused classifiers stay None here too, so AdamW skips them on every rank alike)
                putils.train_step(model, opt, None, config.num_levels, config.top_k_patches, config.task, gb, ar)
                continue
            opt.zero_grad(set_to_none=True)
            outputs, loss = putils.forward_backward(model, batch, config.num_levels, config.top_k_patches, config.task, gb)
            if ar:
                ar(model, num_levels=config.num_levels)
            opt.step()
            n_local = len(batch["slide"])
            train_eval.register(batch, outputs, float(loss.detach()) * gb / n_local, weight=n_local)
        sched.step()
        log(train_eval.calculate(train_stats, e) | {"epoch": e})
        train_eval.reset()
        if e % config.eval_epochs == 0 and val_ds is not None and len(val_ds) > 0:
            _evaluate(model, val_ds, config, val_eval, rank, world)
            d = val_eval.calculate(train_stats, e) | {"epoch": e}
            log(d)
            val_eval.reset()
            val_score = d[f"val_{score_key}"]
            if config.early_stopping and val_score > best_val and e >= config.min_epochs:
                best_val = val_score
                train_stats["epoch"] = e + 1
                if rank == 0:
                    save_state(model_dir, model, train_stats)
                pdist.barrier()
            model.train()
    if config.early_stopping:
        load_state(model_dir, model)
    pdist.barrier()                     # every rank has re-loaded the checkpoint before rank 0 rewrites it (round 4: a rank read
    train_stats["epoch"] = config.num_epochs     # the file while it was being written - the two-rank rehearsal caught it)
    if rank == 0:
        save_state(model_dir, model, train_stats)
    pdist.barrier()
    test_eval = mk_eval("test")
    if test_ds is not None and len(test_ds) > 0:
        _evaluate(model, test_ds, config, test_eval, rank, world)
        final = test_eval.calculate(train_stats) | {"epoch": config.num_epochs}
        log(final)
        train_stats["test"] = final
    return train_stats


def synthetic_dataset(n: int, base_shape, num_levels: int, device, seed: int = 0, first_id: int = 0, nbins: int = 4):
    out = []
    for i in range(first_id, first_id + n):
        s = DeviceSlide.synthetic(seed, i, base_shape, num_levels=num_levels, device=device)
        sb, cen = s.synthetic_spec.label(nbins)
        out.append({"slide": s, "survival_bin": sb, "survival": float(sb) + 0.5, "censored": cen})
    return out


def main():
    import argparse
    from .config import Config
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--model-dir", required=True)
    ap.add_argument("--synthetic", type=int, default=0, help="train on N synthetic slides (8x8 level-0 grid) instead of real data")
    args = ap.parse_args()
    config = Config.load(args.model_dir, test_mode=True)
    rank, world, local = pdist.env_rank_world()
    torch.manual_seed(config.seed)
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    dev = torch.device("cuda", torch.cuda.current_device())
    pdist.init(os.environ.get("PATHS_DIST_BACKEND", "nccl"), dev)
    model = config.get_model().to(dev)
    if not args.synthetic:
        raise SystemExit("real-data loading (CSV labels + split files) is outside this build's scope: build DeviceSlide objects with "
                         "DeviceSlide.from_preprocessed(...) and call paths_amd.train.train_loop")
    n = args.synthetic
    ds = synthetic_dataset(n, (8, 8), config.num_levels, dev)
    k = max(1, n // 5)
    train_loop(model, ds[2 * k:], ds[:k], ds[k:2 * k], config, args.model_dir)


if __name__ == "__main__":
    main()
