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


def gener_target_pseudo(model, images, names, save_pseudo_label_path, num_classes, slide=True, save_prob=True,
                        size=None, cutoff_top=0.8, cutoff_low=0.6, ignore_label=-1, save_dtype=torch.float32):
    """Offline pseudo-label generation (pseudo_generation.py:96-155): eval-mode sliding-window forward with the
    8-way TTA, then `<fname>.pt` = torch.save of the (C,H,W) fp32 probability map (the wire format
    `BaseData.__getitem__` loads, basedata.py:87).  `images` yields (1,3,H,W) CUDA tensors, `names` the file names.
    `save_dtype=torch.float16` halves the files (SURVEY 8 f1 option; `load_target_pseudo` returns fp32 either way;
    the reference's loader would need `.float()` for such a file).
    Returns the hard labels selected from each map (what the reference only renders as colour PNGs)."""
    import os
    from ..utils.tools import pre_slide
    os.makedirs(save_pseudo_label_path, exist_ok=True)
    model.eval()
    hards = []
    with torch.no_grad():
        for img, name in zip(images, names):
            cls = pre_slide(model, img, num_classes=num_classes, tta=True) if slide else model(img)
            if size is not None and tuple(size) != tuple(cls.shape[-2:]):
                raise UemError("gener_target_pseudo: resizing to a different `size` is not implemented "
                               "(the ISPRS / LoveDA tiles are generated at their native size)")
            if save_prob:
                torch.save(cls.squeeze(dim=0).to("cpu", save_dtype), os.path.join(save_pseudo_label_path, name + '.pt'))
            hards.append(pseudo_selection(cls, cutoff_top, cutoff_low, 'tensor', ignore_label))
    return hards


def load_target_pseudo(path, device="cuda"):
    """`<fname>.pt` -> (C,H,W) fp32 soft pseudo label on `device` (basedata.py:87 reads the same file with torch.load)."""
    t = torch.load(path, map_location="cpu")
    if t.dim() != 3:
        raise UemError(f"{path}: expected a (C,H,W) probability map, got shape {tuple(t.shape)}")
    return t.float().to(device)
