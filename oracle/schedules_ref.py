"""LR schedules of the two training loops, restated (TEST INFRASTRUCTURE, see oracle/__init__.py).

  src/ssl4polyp/models/mae/util/lr_sched.py:9-21            per-iteration half-cosine with linear warm-up from 0
  src/ssl4polyp/classification/train_classification.py:3952-3957   per-epoch cosine lambda, warm-up (e+1)/W
"""
import math


def mae_lr(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """lr_sched.py:11-15 (``epoch`` is fractional: data_iter_step / len(loader) + epoch)."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def cls_cosine_lambda(epoch: int, warmup_epochs: int, total_epochs: int) -> float:
    """tc.py:3952-3957."""
    if warmup_epochs > 0 and epoch < warmup_epochs:
        return float(epoch + 1) / float(max(1, warmup_epochs))
    progress = (epoch - warmup_epochs) / float(max(1, total_epochs - warmup_epochs))
    progress = min(max(progress, 0.0), 1.0)
    return 0.5 * (1.0 + math.cos(math.pi * progress))
