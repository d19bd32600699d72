"""Evaluation on the MI355X (reference uemda/utils/eval.py:14-56 + uemda/gast/metrics.py:19-65): eval-mode forward
(optionally sliding window / TTA), argmax and confusion matrix on the device, IoU / F1 / precision / recall per
class on the host.  The per-class formulas are `ever.api.metric.pixel.PixelMetric`'s (third-party, absent from the
reference tree: restated from their standard definitions)."""
import numpy as np
import torch

from .. import ops
from .tools import pre_slide


class ConfusionMeter:
    def __init__(self, num_classes, ignore_labels=(), device="cuda"):
        self.num_classes = num_classes
        self.ignore_labels = sorted(ignore_labels, reverse=True)
        self.cm = torch.zeros((num_classes, num_classes), device=device, dtype=torch.int64)

    def update(self, prob, gt):
        """prob (B,C,H,W) float32 class scores, gt (B,H,W) int64 with negative = unlabeled (eval.py:41-47)."""
        prob, gt = prob.contiguous(), gt.contiguous().long()
        B, C, H, W = prob.shape
        pred = torch.empty((B, H, W), device=prob.device, dtype=torch.int64)
        ops.call("uem_argmax_confusion", ops.ptr(prob), ops.ptr(gt), ops.ptr(pred), ops.ptr(self.cm), B, C, H * W, ops.stream())
        return pred

    def summary(self, dec=5):
        cm = self.cm.cpu().numpy().astype(np.float64)            # rows = ground truth, cols = prediction
        tp = np.diag(cm)
        gt_n, pred_n = cm.sum(1), cm.sum(0)
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = tp / (gt_n + pred_n - tp)
            precision, recall = tp / pred_n, tp / gt_n
            f1 = 2 * precision * recall / (precision + recall)
        keep = [i for i in range(self.num_classes) if i not in self.ignore_labels]     # metrics.py:36-42
        out = {k: np.round(v[keep], dec) for k, v in dict(iou=iou, f1=f1, precision=precision, recall=recall).items()}
        out.update(miou=float(np.round(out["iou"].mean(), dec)), mf1=float(np.round(out["f1"].mean(), dec)),
                   confusion=cm)
        return out


def evaluate(model, batches, num_classes, ignore_labels=(), slide=True, tta=False):
    """`batches` yields (image (B,3,H,W), label (B,H,W)) CUDA tensors; returns the metric dict and mIoU."""
    model.eval()
    meter = ConfusionMeter(num_classes, ignore_labels)
    with torch.no_grad():
        for image, label in batches:
            cls = pre_slide(model, image, num_classes=num_classes, tta=tta) if slide else model(image)
            meter.update(cls, label)
    res = meter.summary()
    return res, res["miou"]
