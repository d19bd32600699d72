"""Oracle restatement of the inference-side rows.  TEST INFRASTRUCTURE ONLY.
  pre_slide    reference uemda/utils/tools.py:61-97 (pinned by tests/golden/pre_slide.npz)
  tta_predict  reference tools.py:132-152 over `ttach` (third-party, absent: restated from its documented
               HorizontalFlip / Rotate90 semantics -- parity unpinned)
  pseudo_prob_map   the `<fname>.pt` tensor of gener_target_pseudo, uemda/gast/pseudo_generation.py:128-136 (pinned, slide=False leg)
  evaluate_pairs    what evaluate feeds the metric, uemda/utils/eval.py:39-47 (pinned by tests/golden/evaluate_pairs.npz)
  confusion / per-class metrics: uemda/utils/eval.py:41-50 + ever's PixelMetric (third-party, absent: formulas unpinned)."""
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


def pseudo_prob_map(model, image, size, num_classes, slide=True, tta=True):
    """The tensor gener_target_pseudo writes as `<fname>.pt` with save_prob=True (uemda/gast/pseudo_generation.py:128-136): the
    model's (sliding-window) probability map resized to `size` with bilinear align_corners=True, batch dimension squeezed.
    Pinned by tests/golden/gener_pseudo.npz on the slide=False leg (the TTA leg needs `ttach`: unpinned)."""
    with torch.no_grad():
        cls = pre_slide(model, image, num_classes=num_classes, tta=tta) if slide else model(image)
        return F.interpolate(cls, size, mode="bilinear", align_corners=True).squeeze(dim=0)


def evaluate_pairs(model, image, gt, num_classes, slide=True, tta=False, tile_size=(512, 512)):
    """What `evaluate` hands the metric for one batch (uemda/utils/eval.py:39-47): argmax of the (sliding-window) map and the labels,
    both restricted to pixels with label >= 0 -> (y_true, y_pred) int arrays.  Pinned by tests/golden/evaluate_pairs.npz."""
    with torch.no_grad():
        cls = pre_slide(model, image, num_classes=num_classes, tile_size=tile_size, tta=tta) if slide else model(image)
    pred = cls.argmax(dim=1).cpu().numpy()
    g = gt.cpu().numpy().astype(np.int32)
    mask = g >= 0
    return g[mask].ravel(), pred[mask].ravel()


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
