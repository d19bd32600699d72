"""Host-side helpers of the training scripts that sit on the path (reference uemda/utils/tools.py):
loss_calc (:240-260), lr_poly / lr_warmup / adjust_learning_rate (:191-207), seed_torch (:305-314)."""
import argparse
import os
import random

import numpy as np
import torch

from ..gast.balance import CrossEntropy
from ..ops import UemError


def str2bool(v):
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Unsupported value encountered.')


def lr_poly(base_lr, i_iter, max_iter, power):
    return base_lr * ((1 - float(i_iter) / max_iter) ** power)


def lr_warmup(base_lr, i_iter, warmup_iter):
    return base_lr * (float(i_iter) / warmup_iter)


def adjust_learning_rate(optimizer, i_iter, cfg):
    if i_iter < cfg.PREHEAT_STEPS:
        lr = lr_warmup(cfg.LEARNING_RATE, i_iter, cfg.PREHEAT_STEPS)
    else:
        lr = lr_poly(cfg.LEARNING_RATE, i_iter, cfg.NUM_STEPS, cfg.POWER)
    optimizer.param_groups[0]['lr'] = lr
    if len(optimizer.param_groups) > 1:
        optimizer.param_groups[1]['lr'] = lr * 10
    return lr


def loss_calc(pred, label, loss_fn, multi=False):
    """Cross-entropy of (a list of) low-resolution logits against a full-resolution label map.  The
    align_corners=True upsample of the reference (tools.py:249-250) happens inside the fused loss kernel."""
    if not isinstance(loss_fn, CrossEntropy):
        raise UemError("loss_calc: loss_fn must be uemda_amd.gast.balance.CrossEntropy")
    if multi is True:
        if len(pred) > 2:
            raise UemError("loss_calc: at most two heads")
        return loss_fn.forward_multi(list(pred), label.long())
    return loss_fn(pred, label.long())


def seed_torch(seed=2333):
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
