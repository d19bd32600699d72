"""Oracle restatement of one training iteration.  TEST INFRASTRUCTURE ONLY.

  ssl_step : reference tools/train_ssl_uem.py:193-232 (2 forwards, label_refine, pseudo_selection,
             prototype EMA, CE + UVEM losses, backward, clip_grad_norm_(32), SGD(momentum, wd))
  src_step : reference tools/train_src.py:112-141 (forward, CE, backward, clip, SGD)
"""
import torch

from . import gast

HYPER = dict(lr=1e-2, momentum=0.9, weight_decay=5e-4, max_norm=32.0, cutoff_top=0.8,
             cutoff_low=0.6, refine_mode="all", refine_temp=2.0, uvem_m=0.2, uvem_t=0.7,
             uvem_g=4.0, proto_decay=0.996, ignore_label=-1)
# configs/st/uemda/2potsdam.py:9,14,24-25,46-48,59-61 ; train_ssl_uem.py:117 ; configs/ToPotsdam.py


class SGDState:
    """torch.optim.SGD(momentum, weight_decay) restated (no nesterov, no dampening)."""

    def __init__(self, params, momentum=0.9, weight_decay=5e-4):
        self.params = list(params)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.buf = [None] * len(self.params)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def clip_grad_norm(self, max_norm):
        grads = [p.grad for p in self.params if p.grad is not None]
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)     # torch.nn.utils.clip_grad_norm_
        for g in grads:
            g.mul_(coef)
        return total

    @torch.no_grad()
    def step(self, lr):
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            d = p.grad + self.weight_decay * p
            if self.buf[i] is None:
                self.buf[i] = d.clone()
            else:
                self.buf[i].mul_(self.momentum).add_(d)
            p.add_(self.buf[i], alpha=-lr)


def ssl_step(model, opt, prototypes, batch, lr, hp=HYPER, dropout=False, n_classes=6):
    model.train()
    ps1, ps2, feat_s = model(batch["images_s"], dropout) if model.use_ppm else model(batch["images_s"])
    pt1, pt2, feat_t = model(batch["images_t"], dropout) if model.use_ppm else model(batch["images_t"])
    with torch.no_grad():
        soft = gast.label_refine(batch["label_t_sup"], feat_t.detach(), [pt1.detach(), pt2.detach()],
                                 batch["label_t_soft"], prototypes, True, hp["refine_mode"],
                                 hp["refine_temp"])
        hard = gast.pseudo_selection(soft, hp["cutoff_top"], hp["cutoff_low"], hp["ignore_label"])
        new_protos, label_ds = gast.update_prototype(feat_s.detach(), batch["label_s"], prototypes,
                                                     n_classes, hp["proto_decay"], hp["ignore_label"])
    loss_s = gast.loss_calc([ps1, ps2], batch["label_s"], hp["ignore_label"])
    loss_t = gast.loss_calc_uvem([pt1, pt2], hard, soft, hp["uvem_m"], hp["uvem_t"], hp["uvem_g"],
                                 hp["ignore_label"], n_classes)
    loss = loss_s + loss_t
    opt.zero_grad()
    loss.backward()
    gnorm = opt.clip_grad_norm(hp["max_norm"])
    opt.step(lr)
    return dict(loss_source=loss_s.detach(), loss_target=loss_t.detach(), label_t_soft=soft,
                label_t_hard=hard, prototypes=new_protos, label_s_ds=label_ds, grad_norm=gnorm,
                pred_t1=pt1.detach(), pred_t2=pt2.detach(), feat_t=feat_t.detach(),
                pred_s1=ps1.detach(), pred_s2=ps2.detach(), feat_s=feat_s.detach())


def src_step(model, opt, batch, lr, hp=HYPER, dropout=False, align_domain=False):
    """reference tools/train_src.py:112-141; align_domain = the script's --align-domain (:126-135)."""
    model.train()
    ps1, ps2, feat_s = model(batch["images_s"], dropout) if model.use_ppm else model(batch["images_s"])
    loss_seg = gast.loss_calc([ps1, ps2], batch["label_s"], hp["ignore_label"])
    loss = loss_seg
    out = {}
    if align_domain:
        _p1, _p2, feat_t = model(batch["images_t"], dropout) if model.use_ppm else model(batch["images_t"])
        k = feat_s.shape[1]
        loss_domain = gast.coral_loss(feat_s.permute(0, 2, 3, 1).reshape(-1, k), feat_t.permute(0, 2, 3, 1).reshape(-1, k))
        loss = loss_seg + loss_domain
        out["loss_domain"] = loss_domain.detach()
    opt.zero_grad()
    loss.backward()
    gnorm = opt.clip_grad_norm(hp["max_norm"])
    opt.step(lr)
    out.update(loss_source=loss_seg.detach(), grad_norm=gnorm, pred_s1=ps1.detach(), pred_s2=ps2.detach())
    return out


def align_step(model, opt, prototypes, batch, lr, hp=HYPER, n_classes=6, align_domain=True, pcl_temp=8.0):
    """One stage-2 iteration: reference tools/train_align_uem.py:139-183."""
    import torch.nn.functional as F
    model.train()
    ps1, ps2, feat_s = model(batch["images_s"])
    with torch.no_grad():
        new_protos, label_s_down = gast.update_prototype(feat_s.detach(), batch["label_s"], prototypes, n_classes,
                                                         hp["proto_decay"], hp["ignore_label"])
    pt1, pt2, feat_t = model(batch["images_t"])
    with torch.no_grad():
        size = batch["images_t"].shape[-2:]
        x1 = F.interpolate(pt1, size, mode="bilinear", align_corners=True)
        x2 = F.interpolate(pt2, size, mode="bilinear", align_corners=True)
        soft = (x1.softmax(1) + x2.softmax(1)) * 0.5
        soft = gast.label_refine(batch["label_t_sup"], feat_t.detach(), [pt1.detach(), pt2.detach()], soft, new_protos,
                                 True, hp["refine_mode"], hp["refine_temp"])
        hard = gast.pseudo_selection(soft, hp["cutoff_top"], hp["cutoff_low"], hp["ignore_label"])
        label_t = gast.downscale_label(hard, n_classes, 16, hp["ignore_label"], 0.75)
    loss_seg = gast.loss_calc([ps1, ps2], batch["label_s"], hp["ignore_label"])
    k = feat_s.shape[1]
    loss_domain = gast.coral_loss(feat_s.permute(0, 2, 3, 1).reshape(-1, k), feat_t.permute(0, 2, 3, 1).reshape(-1, k)) \
        if align_domain else 0.0
    loss_align = 0.5 * (gast.pcl_loss(new_protos, feat_s, label_s_down, pcl_temp, hp["ignore_label"]) +
                        gast.pcl_loss(new_protos, feat_t, label_t, pcl_temp, hp["ignore_label"]))
    loss = loss_seg + loss_domain + loss_align
    opt.zero_grad()
    loss.backward()
    gnorm = opt.clip_grad_norm(hp["max_norm"])
    opt.step(lr)
    return dict(loss_seg=loss_seg.detach(), loss_domain=torch.as_tensor(loss_domain).detach(), loss_align=loss_align.detach(),
                prototypes=new_protos, label_t_hard=hard, grad_norm=gnorm, pred_s1=ps1.detach(), pred_t1=pt1.detach())
