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


# ---- sliding-window / test-time-augmentation inference (reference tools.py:53-97,132-152) --------------------
def pad_image(img, target_size):
    """tools.py:53-58, verbatim semantics: F.pad(img, (0, 0, rows_missing, cols_missing)) -- i.e. the reference
    pads only the HEIGHT (top by rows_missing, bottom by cols_missing); it is a no-op for every window pre_slide
    cuts from an image at least as large as the tile.  Pure data movement."""
    import torch.nn.functional as tnf
    rows_missing, cols_missing = target_size[0] - img.shape[2], target_size[1] - img.shape[3]
    if rows_missing == 0 and cols_missing == 0:
        return img
    return tnf.pad(img, (0, 0, rows_missing, cols_missing), 'constant', 0)


def tta_predict(model, img, batched=True):
    """hflip x rot90{0,90,180,270}: 8 eval-mode views, de-augmented and averaged (tools.py:132-152; ttach semantics:
    augment = hflip then rot90(k), de-augment = rot90(-k) then hflip).  Like the reference, the mean is taken over
    the concatenated batch dimension, so it is only meaningful for batch size 1.

    `batched` (SURVEY 8 f1, D4-symmetry batching): views of equal shape go through the network as ONE batch -- all 8
    for a square tile, 4 + 4 otherwise -- instead of 8 single-image forwards that cannot fill the device (a 512x512
    tile is 8 row tiles of the layer-4 GEMMs).  Eval-mode outputs do not depend on the batch they travel in and
    the de-augmented views are summed in the same order, so the result is the same as the sequential form."""
    from .. import ops
    if img.shape[0] != 1:
        raise UemError("tta_predict: the reference averages over cat(xs, 0); use batch size 1")
    combos = [(flip, k) for flip in (False, True) for k in range(4)]
    views = [torch.rot90(img.flip(3) if flip else img, k, (2, 3)).contiguous() for flip, k in combos]
    outs = [None] * 8
    if batched and not getattr(model, "training", False):
        groups = {}
        for i, v in enumerate(views):
            groups.setdefault(tuple(v.shape[2:]), []).append(i)
        for idx in groups.values():
            y = model(torch.cat([views[i] for i in idx], 0))
            for j, i in enumerate(idx):
                outs[i] = y[j:j + 1]
    else:
        outs = [model(v) for v in views]
    acc = None
    for (flip, k), x in zip(combos, outs):
        x = torch.rot90(x, -k, (2, 3))
        x = (x.flip(3) if flip else x).contiguous()
        acc = x.clone() if acc is None else ops.add_(acc, x)
    ops.call("uem_scale", ops.ptr(acc), acc.numel(), 1.0 / 8.0, ops.stream())
    return acc


def pre_slide(model, image, num_classes=7, tile_size=(512, 512), tta=False):
    """overlap-averaged sliding-window inference, tile 512 / stride 256 (tools.py:61-97)."""
    from math import ceil
    from .. import ops
    ops.need_gpu(image)
    B, _, H, W = image.shape
    stride = ceil(tile_size[0] * (1 - 1 / 2))
    tile_rows = int(ceil((H - tile_size[0]) / stride) + 1)
    tile_cols = int(ceil((W - tile_size[1]) / stride) + 1)
    full = torch.zeros((B, num_classes, H, W), device=image.device, dtype=torch.float32)
    cnt = torch.zeros((B, 1, H, W), device=image.device, dtype=torch.float32)
    for row in range(tile_rows):
        for col in range(tile_cols):
            x1, y1 = int(col * stride), int(row * stride)
            x2, y2 = min(x1 + tile_size[1], W), min(y1 + tile_size[0], H)
            x1, y1 = max(int(x2 - tile_size[1]), 0), max(int(y2 - tile_size[0]), 0)
            img = image[:, :, y1:y2, x1:x2]
            padded_img = pad_image(img, tile_size).contiguous()
            padded = tta_predict(model, padded_img) if tta else model(padded_img)
            padded = padded.contiguous()
            ops.call("uem_window_accumulate", ops.ptr(full), ops.ptr(cnt), ops.ptr(padded), B, num_classes, H, W, y1, x1,
                     y2 - y1, x2 - x1, padded.shape[2], padded.shape[3], ops.stream())
    ops.call("uem_window_normalize", ops.ptr(full), ops.ptr(cnt), B, num_classes, H, W, ops.stream())
    return full
