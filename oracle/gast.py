"""Oracle (CPU, fp32) restatement of the pseudo-label mining path.  TEST INFRASTRUCTURE ONLY.

Each function cites the reference lines it restates.  Shapes: B images, C classes, k feature
channels, (h, w) feature map, (H, W) = 16*(h, w) tile.
"""
import math

import torch
import torch.nn.functional as F

EPS = 1e-7


# ------------------------------------------------------------------------------------------------
# torch_scatter.scatter  (third-party, pytorch-scatter 2.0.x; call sites alignment.py:187,245)
# ------------------------------------------------------------------------------------------------
def scatter(src, index, dim=1, reduce="max", dim_size=None):
    """out[b, s, c] = reduce_{p : index[b, p] == s} src[b, p, c]; untouched segments -> 0.

    Published semantics of torch_scatter.scatter: `index` broadcasts against `src`, the output
    size along `dim` is index.max()+1, 'max'/'min' return values only and fill segments that
    receive nothing with 0.  "parity unpinned" at this boundary (library absent).
    """
    assert reduce in ("max", "sum", "mean", "min")
    index = index.expand_as(src) if index.shape != src.shape else index
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    out = torch.zeros(shape, dtype=src.dtype)
    red = {"max": "amax", "min": "amin", "sum": "sum", "mean": "mean"}[reduce]
    return out.scatter_reduce(dim, index, src, reduce=red, include_self=False)


# ------------------------------------------------------------------------------------------------
# Pearson distance, GEMM form  (alignment.py:424-451)
# ------------------------------------------------------------------------------------------------
def pearson_dist(x, protos):
    """d[n, m] = 0.5 * (1 - cov/(k-1+eps) / (std_x*std_p + eps)), std unbiased."""
    k = x.shape[-1]
    xc = x - x.mean(dim=-1, keepdim=True)
    pc = protos - protos.mean(dim=-1, keepdim=True)
    cov = (xc @ pc.t()) / (k - 1 + EPS)
    div = x.std(dim=-1).unsqueeze(1) * protos.std(dim=-1).unsqueeze(0)
    return (1.0 - cov / (div + EPS)) * 0.5


def _max_norm(p):
    return p / (p.max(dim=1, keepdim=True)[0] + 1e-7)


def _up(x, size):
    return F.interpolate(x, size, mode="bilinear", align_corners=True)


# ------------------------------------------------------------------------------------------------
# Aligner.label_refine  (alignment.py:194-293)
# ------------------------------------------------------------------------------------------------
def label_refine(label_t_sup, feat_t, preds_t, label_t_soft, prototypes, refine=True, mode="all",
                 temp=2.0, sup_ignore_id=None):
    """Three-view refinement of the soft pseudo label.  `sup_ignore_id=None` reproduces the
    reference's batch-global `label_t_sup.max()` (alignment.py:241)."""
    assert mode in ("all", "s", "p", "l")
    if not refine:
        return label_t_soft
    b, k, h, w = feat_t.shape
    C = label_t_soft.shape[1]
    H, W = label_t_soft.shape[-2:]
    weight = None
    if mode in ("all", "p"):                                          # prototype view :215-223
        flat = feat_t.permute(0, 2, 3, 1).reshape(-1, k)
        sim = 1.0 / pearson_dist(flat, prototypes)
        sim = sim.view(b, h, w, C).permute(0, 3, 1, 2)
        pw = _max_norm(torch.softmax(_up(sim, (H, W)), dim=1))
        weight = pw
    if mode in ("all", "l"):                                          # prediction view :225-236
        if isinstance(preds_t, (list, tuple)):
            lw = 0.5 * (torch.softmax(_up(preds_t[0], (H, W)) / temp, dim=1) +
                        torch.softmax(_up(preds_t[1], (H, W)) / temp, dim=1))
        else:
            lw = torch.softmax(_up(preds_t, (H, W)) / temp, dim=1)
        lw = _max_norm(lw)
        weight = lw if weight is None else weight + lw
    if mode in ("all", "s"):                                          # superpixel view :238-258
        sup = label_t_sup.reshape(b, -1, 1)
        ign_id = sup.max() if sup_ignore_id is None else sup_ignore_id
        ignored = (sup == ign_id).reshape(b, 1, H, W)
        soft_flat = label_t_soft.permute(0, 2, 3, 1).reshape(b, -1, C)
        seg = scatter(soft_flat, sup, dim=1, reduce="max")
        pix = torch.gather(seg, 1, sup.expand(-1, -1, C)).reshape(b, H, W, C).permute(0, 3, 1, 2)
        sw = _max_norm(torch.softmax(pix / temp, dim=1))
        if mode == "all":
            weight = torch.where(ignored, weight, weight * sw)
        else:
            weight = torch.where(ignored, torch.ones_like(sw), sw)
    out = weight * label_t_soft
    return out / (out.sum(dim=1, keepdim=True) + EPS)                 # _logits_norm :316-326


# ------------------------------------------------------------------------------------------------
# pseudo_selection  (pseudo_generation.py:59-93)
# ------------------------------------------------------------------------------------------------
def pseudo_selection(mask, cutoff_top=0.8, cutoff_low=0.6, ignore_label=-1):
    b, c, h, w = mask.shape
    m = mask.reshape(b, c, -1)
    thr = m.max(dim=-1, keepdim=True)[0] * cutoff_top
    thr = torch.maximum(thr, torch.tensor(cutoff_low, dtype=mask.dtype))
    g = m > thr
    cnt = g.sum(dim=1)
    lab = g.to(torch.uint8).argmax(dim=1)
    lab = torch.where(cnt == 1, lab, torch.full_like(lab, ignore_label))
    return lab.view(b, h, w)


# ------------------------------------------------------------------------------------------------
# DownscaleLabel  (alignment.py:484-509)
# ------------------------------------------------------------------------------------------------
def downscale_label(label, n_classes, scale=16, ignore_label=-1, min_ratio=0.75):
    if label.dim() == 4:
        label = label.squeeze(1)
    b, H, W = label.shape
    lab = torch.where(label == ignore_label, torch.full_like(label, n_classes), label)
    cnt = torch.zeros(b, n_classes + 1, H // scale, W // scale, dtype=torch.float32)
    blocks = lab.reshape(b, H // scale, scale, W // scale, scale).permute(0, 1, 3, 2, 4).reshape(
        b, H // scale, W // scale, scale * scale)
    for c in range(n_classes + 1):
        cnt[:, c] = (blocks == c).sum(-1).float()
    ratio, idx = (cnt / float(scale * scale)).max(dim=1, keepdim=True)
    idx = torch.where(idx == n_classes, torch.full_like(idx, ignore_label), idx)
    idx = torch.where(ratio < min_ratio, torch.full_like(idx, ignore_label), idx)
    return idx                                                         # (b, 1, h, w) int64


# ------------------------------------------------------------------------------------------------
# prototype update  (alignment.py:86-90, 328-355, 463-466)
# ------------------------------------------------------------------------------------------------
def local_prototypes(feat, label_ds, prototypes, n_classes, ignore_label=-1):
    """(sum_c, n_c) -> local[c] = sum_c/(n_c+eps), classes with n_c < 1 keep the old prototype."""
    b, k, h, w = feat.shape
    x = feat.permute(0, 2, 3, 1).reshape(-1, k)
    lab = label_ds.reshape(-1)
    onehot = torch.zeros(lab.numel(), n_classes, dtype=feat.dtype)
    valid = lab != ignore_label
    onehot[valid, lab[valid]] = 1.0
    n_c = onehot.sum(0)
    sums = onehot.t() @ x
    local = sums / (n_c.unsqueeze(1) + EPS)
    return torch.where((n_c < 1).unsqueeze(1), prototypes, local), sums, n_c


def update_prototype(feat, label, prototypes, n_classes, decay=0.996, ignore_label=-1):
    label_ds = downscale_label(label, n_classes, 16, ignore_label, 0.75)
    local, _, _ = local_prototypes(feat.detach(), label_ds, prototypes, n_classes, ignore_label)
    return (1.0 - decay) * local + decay * prototypes, label_ds


def update_avg(feat, label, data_sum, data_cnt, n_classes, ignore_label=-1):
    """Aligner.update_avg (alignment.py:107-120; caller tools/init_prototypes.py:101-109): running per-class feature sums (c, k) and
    pixel counts (c, 1) over the down-scaled labels -> (data_sum', data_cnt').  Pinned by tests/golden/aligner_avg.npz."""
    label_ds = downscale_label(label, n_classes, 16, ignore_label, 0.75)
    _, sums, n_c = local_prototypes(feat.detach(), label_ds, torch.zeros_like(data_sum), n_classes, ignore_label)
    return data_sum + sums, data_cnt + n_c.unsqueeze(1)


def init_avg(data_sum, data_cnt):
    """Aligner.init_avg (alignment.py:122-123): prototypes = sum / (count + eps); an unseen class gets 0 / eps = 0."""
    return data_sum / (data_cnt + EPS)


# ------------------------------------------------------------------------------------------------
# losses  (tools.py:240-260, balance.py:81-101, 345-457)
# ------------------------------------------------------------------------------------------------
def ce_mean_all(logits_full, label, ignore_label=-1, pixel_weight=None):
    """CrossEntropy.forward: per-pixel CE with ignore, then mean over ALL pixels (balance.py:97-101)."""
    loss = F.cross_entropy(logits_full, label, ignore_index=ignore_label, reduction="none").view(-1)
    if pixel_weight is not None:
        loss = loss * pixel_weight
    return loss.mean()


def loss_calc(preds, label, ignore_label=-1, class_balancer=None):
    """loss_calc(multi=True) with CrossEntropy (tools.py:240-254)."""
    total = 0
    for p in preds:
        if p.shape[-2:] != label.shape[-2:]:
            p = _up(p, label.shape[-2:])
        w = class_balancer.get_class_weight_4pixel(label) if class_balancer is not None else None
        total = total + ce_mean_all(p, label.long(), ignore_label, w)
    return total / len(preds)


def uvem_weight(u, m=0.2, t=0.7, gamma=4.0):
    """UVEMLoss.get_weight (balance.py:396-423) in closed form."""
    left = torch.clamp(1.0 - ((u - m) ** 2) * (1.0 / (m ** 2)), 0.0, 1.0) ** (1.0 / gamma) if m > 0 \
        else torch.ones_like(u)
    # reference feeds 1.0 (not u) through the left polynomial where u is outside [0, m]; that
    # branch is never selected by the final where(u <= m) unless u < 0 (unreachable: u >= 0).
    if m < t:
        ur = torch.where((u > m) & (u <= t), u, torch.zeros_like(u))
        right = torch.clamp(1.0 - ((ur - m) ** 2) * (1.0 / ((t - m) ** 2)), 0.0, 1.0) ** (1.0 / gamma)
    else:
        right = torch.zeros_like(u)
    wgt = torch.where(u <= m, left, right)
    return torch.where(u >= t, torch.zeros_like(u), wgt)


def uvem_loss(logits_full, hard, soft, m=0.2, t=0.7, gamma=4.0, ignore_label=-1, n_classes=6):
    """UVEMLoss.forward (balance.py:356-394) for one head."""
    ce = F.cross_entropy(logits_full, hard, ignore_index=ignore_label, reduction="none").view(-1)
    lts = soft.permute(0, 2, 3, 1).reshape(-1, n_classes)
    u = torch.sum(-lts * torch.log(lts), dim=1).detach()
    ce = torch.where(u > t, torch.zeros_like(ce), ce)
    w = uvem_weight(u, m, t, gamma)
    valid = ((u <= t) & (hard.view(-1) != ignore_label)).sum()
    return (w * ce).sum() / (valid + 1e-7)


def loss_calc_uvem(preds, hard, soft, m=0.2, t=0.7, gamma=4.0, ignore_label=-1, n_classes=6):
    total = 0                                                          # balance.py:437-451
    for p in preds:
        if p.shape[-2:] != hard.shape[-2:]:
            p = _up(p, hard.shape[-2:])
        total = total + uvem_loss(p, hard.long(), soft, m, t, gamma, ignore_label, n_classes)
    return total / len(preds)


def pcl_loss(protos, feat, labels, temperature=8.0, ignore_label=-1):
    """PrototypeContrastiveLoss.forward (uemda/loss.py:18-47)."""
    if feat.dim() != 2:
        feat = feat.permute(0, 2, 3, 1).reshape(-1, feat.size(1))
    labels = labels.reshape(-1)
    mask = labels != ignore_label
    f = F.normalize(feat[mask], p=2, dim=1)
    p = F.normalize(protos, p=2, dim=1)
    return F.cross_entropy(f @ p.t() / temperature, labels[mask])


def coral_loss(source, target):
    """CoralLoss.forward (uemda/gast/coral.py:27-47)."""
    d = source.shape[1]
    xm = source.mean(0, keepdim=True) - source
    xc = xm.t() @ xm / (source.shape[0] - 1)
    xmt = target.mean(0, keepdim=True) - target
    xct = xmt.t() @ xmt / (target.shape[0] - 1)
    return ((xc - xct) ** 2).sum() / (4 * d * d)


class ClassBalance:
    """balance.py:15-78 restated (EMA of class frequency -> per-pixel weight)."""

    def __init__(self, class_num=6, ignore_label=-1, decay=0.99, temperature=0.5):
        self.class_num, self.ignore_label = class_num, ignore_label
        self.decay, self.temperature, self.eps = decay, temperature, 1e-7
        self.freq = torch.ones(class_num) / class_num

    def _counts(self, label):
        lab = label.reshape(-1)
        valid = lab != self.ignore_label
        return torch.bincount(lab[valid], minlength=self.class_num).float(), valid.sum().float()

    def class_weight(self):
        p = torch.softmax((1.0 - self.freq) / self.temperature, dim=0)
        return p / (p.max() + self.eps)

    def get_class_weight_4pixel(self, label):
        cnt, n = self._counts(label)
        self.freq = (1.0 - self.decay) * (cnt / (n + self.eps)) + self.decay * self.freq
        cw = torch.cat([self.class_weight(), torch.zeros(1)])
        lab = label.reshape(-1)
        lab = torch.where(lab == self.ignore_label, torch.full_like(lab, self.class_num), lab)
        return cw[lab]


# ------------------------------------------------------------------------------------------------
# LR schedule  (tools.py:191-207; train_ssl_uem.py:82-84)
# ------------------------------------------------------------------------------------------------
def learning_rate(i_iter, base_lr, preheat_steps, num_steps, power=0.9):
    if i_iter < preheat_steps:
        return base_lr * (float(i_iter) / preheat_steps)
    return base_lr * ((1 - float(i_iter) / num_steps) ** power)


def edge_shrinking(label, win_size=3, region_size=16):
    """reference gast/superpixels.py:129-150 (three nested Python loops) restated with shifted comparisons:
    keep the id where every pixel of the clipped (2*win+1)^2 window equals it, else int(H/region * W/region)."""
    import numpy as np
    lab = np.asarray(label)
    h, w = lab.shape
    cnt_sup = int(h / region_size * w / region_size)
    keep = np.ones((h, w), dtype=bool)
    for dy in range(-win_size, win_size + 1):
        for dx in range(-win_size, win_size + 1):
            ys, ye = max(0, -dy), min(h, h - dy)
            xs, xe = max(0, -dx), min(w, w - dx)
            same = lab[ys:ye, xs:xe] == lab[ys + dy:ye + dy, xs + dx:xe + dx]
            keep[ys:ye, xs:xe] &= same
    return np.where(keep, lab, cnt_sup).astype(lab.dtype)
