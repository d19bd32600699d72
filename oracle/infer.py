"""Oracle restatement of the inference-side rows.  TEST INFRASTRUCTURE ONLY.
  pre_slide    reference uemda/utils/tools.py:61-97 (pinned by tests/golden/pre_slide.npz)
  tta_predict  reference tools.py:132-152 over `ttach` (third-party, absent: restated from its documented
               HorizontalFlip / Rotate90 semantics -- parity unpinned)
  confusion / per-class metrics: uemda/utils/eval.py:41-50 + ever's PixelMetric (third-party, absent)."""
from math import ceil

import numpy as np
import torch
import torch.nn.functional as F


def tta_predict(model, img):
    xs = []
    for flip in (False, True):
        for k in range(4):
            aug = img.flip(3) if flip else img
            aug = torch.rot90(aug, k, (2, 3))
            x = torch.rot90(model(aug), -k, (2, 3))
            xs.append(x.flip(3) if flip else x)
    return torch.mean(torch.cat(xs, 0), dim=0, keepdim=True)


def pre_slide(model, image, num_classes=7, tile_size=(512, 512), tta=False):
    B, _, H, W = image.shape
    stride = ceil(tile_size[0] * 0.5)
    rows = int(ceil((H - tile_size[0]) / stride) + 1)
    cols = int(ceil((W - tile_size[1]) / stride) + 1)
    full = torch.zeros(B, num_classes, H, W)
    cnt = torch.zeros(B, 1, H, W)
    for r in range(rows):
        for c in range(cols):
            x1, y1 = c * stride, r * stride
            x2, y2 = min(x1 + tile_size[1], W), min(y1 + tile_size[0], H)
            x1, y1 = max(x2 - tile_size[1], 0), max(y2 - tile_size[0], 0)
            img = image[:, :, y1:y2, x1:x2]
            img = F.pad(img, (0, 0, tile_size[0] - img.shape[2], tile_size[1] - img.shape[3]))
            out = tta_predict(model, img) if tta else model(img)
            full[:, :, y1:y2, x1:x2] += out[:, :, :y2 - y1, :x2 - x1]
            cnt[:, :, y1:y2, x1:x2] += 1
    return full / cnt


def confusion(prob, gt, num_classes):
    pred = prob.argmax(dim=1).reshape(-1).numpy()
    g = gt.reshape(-1).numpy()
    m = (g >= 0) & (g < num_classes)
    cm = np.zeros((num_classes, num_classes), dtype=np.int64)
    np.add.at(cm, (g[m], pred[m]), 1)
    return cm


def metrics(cm, ignore_labels=()):
    cm = cm.astype(np.float64)
    tp = np.diag(cm)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (cm.sum(1) + cm.sum(0) - tp)
        p, r = tp / cm.sum(0), tp / cm.sum(1)
        f1 = 2 * p * r / (p + r)
    keep = [i for i in range(cm.shape[0]) if i not in ignore_labels]
    return dict(iou=iou[keep], f1=f1[keep], miou=float(iou[keep].mean()))
