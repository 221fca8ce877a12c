"""Step-side host pieces of the reference's ``main()``: the epoch loop with the LR
schedule, early stopping and best-only checkpointing (yolo_toy_regressor_torch.py:392-432), and the quality
log in the record format of ``utils/logger_quality.py:76-113`` (+ optional throughput fields).

Plain host code: no kernels here.  The checkpoint keeps the reference's state-dict key names and dense NCHW
tensors, so ``best_val_iou.pt`` files move between the reference and this package in both directions.
"""
import json
import os
import subprocess
import time
from datetime import datetime
from pathlib import Path

import torch

from .train import EarlyStoppingMax, evaluate, train_one_epoch, yolo_v1_lr_schedule

# utils/logger_quality.py:25-28 — the reference writes to /var/log; here the default is project-local.
LOG_PATH = os.environ.get("QUALITY_LOG_PATH", "logs/yolo_train.log")
SIDE_DIR = Path(os.environ.get("QUALITY_SIDE_DIR", "logs"))


def _git_short():
    """utils/logger_quality.py:30-38."""
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True,
                                       stderr=subprocess.DEVNULL).strip()
    except Exception:
        return "NA"


def is_better(new, old):
    """utils/logger_quality.py:51-70: higher val_iou, then lower val loss, then later epoch."""
    if old is None:
        return True

    def key(r):
        return (float(r.get("val_iou", -1.0)), -float(r.get("val", float("inf"))), int(r.get("epoch", -1)))
    return key(new) > key(old)


def log_quality(epoch, lr, train_loss, val_loss, val_iou, *, best=False, log_path=None, side_dir=None, extra=None):
    """Append one JSON line {"tag","date","commit","epoch","lr","train","val","val_iou"} (same keys, order and
    compact separators as the reference); ``best=True`` tags it QUALITY_BEST and upserts logs/<sha>/best.json iff
    strictly better.  ``extra`` (dict) appends further fields, e.g. images_per_s.  Returns the record."""
    sha = _git_short()
    rec = {"tag": "QUALITY_BEST" if best else "QUALITY", "date": datetime.now().strftime("%Y-%m-%d"), "commit": sha,
           "epoch": int(epoch), "lr": float(lr), "train": float(train_loss), "val": float(val_loss),
           "val_iou": float(val_iou)}
    if extra:
        rec.update(extra)
    path = Path(log_path or LOG_PATH)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "a", encoding="utf-8") as f:
        f.write(json.dumps(rec, separators=(",", ":")) + "\n")
    if best and sha != "NA":
        side = Path(side_dir) if side_dir is not None else SIDE_DIR
        p = side / sha / "best.json"
        old = None
        if p.exists():
            try:
                old = json.loads(p.read_text(encoding="utf-8"))
            except Exception:
                old = None
        if is_better(rec, old):
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text(json.dumps(rec, indent=2, sort_keys=True), encoding="utf-8")
    return rec


def reference_state_dict(model):
    """state_dict with the reference's key names as dense (NCHW-contiguous) CPU tensors: what ``torch.save``
    writes at yolo_toy_regressor_torch.py:421, loadable by the reference's own nn.Conv2d / nn.Linear modules."""
    return {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}


def save_best(model, path):
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(reference_state_dict(model), path)


def load_best(model, path, device=None, opt=None):
    """yolo_toy_regressor_torch.py:430-432.  ``weights_only`` load; refreshes the optimiser's bf16 shadow."""
    sd = torch.load(path, map_location=device or "cpu", weights_only=True)
    model.load_state_dict(sd)
    if opt is not None and hasattr(opt, "sync_shadow"):
        opt.sync_shadow()
    return model


def fit(model, train_loader, val_loader, device, opt, loss_fn, *, total_epochs=135, warmup=5, patience=15,
        max_grad_norm=1.0, best_path="best_val_iou.pt", scaler=None, log_path=None, side_dir=None, verbose=True):
    """The reference's epoch loop (:397-432): per-epoch LR from ``yolo_v1_lr_schedule``, train, evaluate, log,
    save the best-val-IoU weights, stop after ``patience`` epochs without improvement, reload the best weights.
    Returns the history dict of :395 plus images_per_s."""
    early = EarlyStoppingMax(patience=patience)
    history = {"epoch": [], "lr": [], "train_loss": [], "val_loss": [], "val_iou": [], "images_per_s": []}
    best_val_iou = -1.0
    for epoch_num in range(1, total_epochs + 1):
        lr = yolo_v1_lr_schedule(epoch_num - 1, total_epochs=total_epochs, warmup=warmup)
        for pg in opt.param_groups:
            pg["lr"] = lr
        t0 = time.perf_counter()
        train_loss = train_one_epoch(model, train_loader, device, opt, scaler, loss_fn, max_grad_norm)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        seen = getattr(train_one_epoch, "last_count", 0)      # images the epoch just consumed (the loader is not re-iterated)
        ips = seen / dt if dt > 0 else 0.0
        val_loss, val_iou = evaluate(model, val_loader, device, loss_fn)
        if verbose:
            print(f"Epoch {epoch_num:03d} | lr {lr:.6f} | train {train_loss:.4f} | val {val_loss:.4f} | "
                  f"val_iou {val_iou:.4f} | {ips:.0f} img/s")
        for k, v in zip(history, (epoch_num, lr, train_loss, val_loss, val_iou, ips)):
            history[k].append(v)
        extra = {"images_per_s": round(ips, 1)}
        log_quality(epoch_num, lr, train_loss, val_loss, val_iou, log_path=log_path, side_dir=side_dir, extra=extra)
        improved, stop = early.step(val_iou)
        if improved:
            best_val_iou = val_iou
            save_best(model, best_path)
            log_quality(epoch_num, lr, train_loss, val_loss, val_iou, best=True, log_path=log_path, side_dir=side_dir,
                        extra=extra)
        if stop:
            if verbose:
                print("Early stopping triggered.")
            break
    if os.path.exists(best_path):
        load_best(model, best_path, device=device, opt=opt)
    history["best_val_iou"] = best_val_iou
    return history
