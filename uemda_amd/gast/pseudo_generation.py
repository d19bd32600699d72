"""pseudo_selection on the MI355X (reference uemda/gast/pseudo_generation.py:59-93)."""
import torch

from .. import ops
from ..ops import UemError, call, ptr, stream


def _select(mask, plane_max, cutoff_top, cutoff_low, ignore_label, check_range=True):
    B, C, H, W = mask.shape
    hard = torch.empty((B, H, W), device=mask.device, dtype=torch.int64)
    flag = torch.zeros((), device=mask.device, dtype=torch.int32)
    call("uem_pseudo_select", ptr(mask), ptr(plane_max), ptr(hard), ptr(flag), B, C, H * W, float(cutoff_top),
         float(cutoff_low), int(ignore_label), stream())
    if check_range and int(flag.item()) != 0:
        # the reference asserts 0 <= mask <= 1 (pseudo_generation.py:71), which is a host sync there too
        raise AssertionError("pseudo_selection: mask values outside [0, 1]")
    return hard


def pseudo_selection(mask, cutoff_top=0.8, cutoff_low=0.6, return_type='ndarray', ignore_label=-1,
                     _plane_max=None, check_range=True):
    """(b, c, h, w) probabilities -> (b, h, w) int64 labels: the unique class above its per-image,
    per-class threshold max(cutoff_top * max_p, cutoff_low), else `ignore_label`."""
    assert return_type in ['ndarray', 'tensor']
    ops.need_gpu(mask)
    if mask.dim() != 4 or mask.dtype != torch.float32:
        raise UemError("pseudo_selection expects a float32 (b, c, h, w) tensor")
    m = mask.contiguous()
    B, C, H, W = m.shape
    if _plane_max is None:
        _plane_max = torch.empty((B, C), device=m.device, dtype=torch.int32)
        call("uem_plane_max", ptr(m), ptr(_plane_max), B, C, H * W, stream())
    ret = _select(m, _plane_max, cutoff_top, cutoff_low, ignore_label, check_range)
    return ret.cpu().numpy() if return_type == 'ndarray' else ret


# What `eval(_cfg.DATASETS)` resolves to in the reference (uemda/datasets/isprsda.py:17-36, loveda.py:17-41): the class
# count, the palette of the colour previews and the native tile size -- data constants, restated here because the dataset
# classes themselves (file loaders) are outside the path.
DATASET_INFO = {
    "IsprsDA": dict(num_classes=6, size=(512, 512),
                    palette=[255, 0, 0, 255, 255, 255, 0, 0, 255, 0, 255, 255, 0, 255, 0, 255, 255, 0]),
    "LoveDA": dict(num_classes=7, size=(1024, 1024),
                   palette=[255, 255, 255, 255, 0, 0, 255, 255, 0, 0, 0, 255, 159, 129, 183, 0, 255, 0, 255, 195, 128]),
}


def _dataset_info(_cfg, model):
    name = getattr(_cfg, "DATASETS", None)
    if isinstance(name, str) and name in DATASET_INFO:
        return DATASET_INFO[name]
    if hasattr(name, "LABEL_MAP"):                                  # the dataset class itself
        return dict(num_classes=len(name.LABEL_MAP), palette=getattr(name, "PALETTE", None), size=getattr(name, "SIZE", None))
    n = getattr(_cfg, "NUM_CLASSES", None) or getattr(getattr(model, "config", None), "num_classes", None)
    if n is None:
        raise UemError(f"gener_target_pseudo: cannot tell the class count of _cfg.DATASETS={name!r}")
    return dict(num_classes=int(n), palette=None, size=None)


def _resize_align_corners(cls, size):
    """F.interpolate(cls, size, mode='bilinear', align_corners=True) (pseudo_generation.py:135) on the device: every (b, c)
    plane is a one-channel NHWC image for uem_bilinear_up_fwd."""
    B, C, h, w = cls.shape
    H, W = int(size[0]), int(size[1])
    if (H, W) == (h, w):
        return cls
    src = cls.contiguous()
    out = torch.empty((B, C, H, W), device=cls.device, dtype=torch.float32)
    call("uem_bilinear_up_fwd", ptr(src), ptr(out), B * C, h, w, 1, H, W, 1, 1, None, None, 0, stream())
    return out


def _save_indexed_png(arr, path, palette):
    from PIL import Image                                           # the reference's VisualizeSegmm (uemda/viz.py:11-28)
    import numpy as np
    im = Image.fromarray(np.asarray(arr).astype(np.uint8).squeeze())
    if palette is not None:
        im.putpalette(palette)
    im.save(path)


def gener_target_pseudo(_cfg, model, pseudo_loader, save_pseudo_label_path, slide=True, save_prob=False, size=(1024, 1024),
                        ignore_label=-1, save_dtype=torch.float32):
    """Offline pseudo-label generation with the reference's signature and behaviour (uemda/gast/pseudo_generation.py:96-155;
    caller tools/train_ssl_uem.py:177-190): eval-mode sliding-window forward with the 8-way TTA over `pseudo_loader`
    (batches `(image (b,3,H,W), {'fname': [...]})`), then per image
      save_prob=True : `<fname>.pt` = torch.save of the (C, *size) probability map, resized with bilinear
                       align_corners=True when `size` differs from the image (the wire format BaseData.__getitem__ reads
                       back, basedata.py:87); colour previews of the selected labels when _cfg.SNAPSHOT_DIR is set;
      save_prob=False: `<fname>` = uint8 image of class id + 1 (0 = ignored), from pseudo_selection when
                       _cfg.PSEUDO_SELECT else argmax.
    `save_dtype=torch.float16` (not in the reference) halves the .pt files; `load_target_pseudo` returns fp32 either way."""
    import os
    import numpy as np
    from ..utils.tools import pre_slide
    model.eval()
    info = _dataset_info(_cfg, model)
    num_classes, palette = info["num_classes"], info["palette"]
    color_dir = save_pseudo_label_path + '_color'
    os.makedirs(save_pseudo_label_path, exist_ok=True)
    os.makedirs(color_dir, exist_ok=True)
    snapshot = getattr(_cfg, "SNAPSHOT_DIR", None) is not None
    with torch.no_grad():
        for ret, ret_gt in pseudo_loader:
            ret = ret.cuda()
            cls = pre_slide(model, ret, num_classes=num_classes, tta=True) if slide else model(ret)      # (b, c, h, w)
            names = ret_gt['fname']
            if save_prob:
                full = _resize_align_corners(cls, size)
                torch.save(full.squeeze(dim=0).to("cpu", save_dtype), os.path.join(save_pseudo_label_path, names[0] + '.pt'))
                if snapshot:
                    hard = pseudo_selection(cls, ignore_label=ignore_label, cutoff_top=_cfg.CUTOFF_TOP, cutoff_low=_cfg.CUTOFF_LOW)
                    for fname, pred in zip(names, hard):
                        _save_indexed_png(pred, os.path.join(color_dir, fname.replace('.tif', '.png')), palette)
            else:
                if getattr(_cfg, "PSEUDO_SELECT", False):
                    hard = pseudo_selection(cls, ignore_label=ignore_label)                              # (b, h, w) in -1..C-1
                else:
                    pred = torch.empty((cls.shape[0],) + tuple(cls.shape[-2:]), device=cls.device, dtype=torch.int64)
                    call("uem_argmax_confusion", ptr(cls.contiguous()), None, ptr(pred), None, cls.shape[0], cls.shape[1],
                         cls.shape[2] * cls.shape[3], stream())
                    hard = pred.cpu().numpy()
                _save_indexed_png((hard + 1).reshape(*size), os.path.join(save_pseudo_label_path, names[0]), None)
                if snapshot:
                    for fname, pred in zip(names, hard):
                        _save_indexed_png(pred, os.path.join(color_dir, fname.replace('.tif', '.png')), palette)


def load_target_pseudo(path, device="cuda"):
    """`<fname>.pt` -> (C,H,W) fp32 soft pseudo label on `device` (basedata.py:87 reads the same file with torch.load)."""
    t = torch.load(path, map_location="cpu")
    if t.dim() != 3:
        raise UemError(f"{path}: expected a (C,H,W) probability map, got shape {tuple(t.shape)}")
    return t.float().to(device)
