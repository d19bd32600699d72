"""Superpixel label maps on either side of the path (reference uemda/gast/superpixels.py, datasets/basedata.py:77-79).

The LSC segmentation itself is OpenCV preprocessing and stays outside (out of scope); what is here is the
part that feeds `Aligner.label_refine`: the edge shrinking that marks a 7x7-unanimous core of every superpixel and
sends the rest to the ignored id, and the on-disk format, `<name>.tif` holding one int32 id per pixel
(`skimage.io.imsave` = uncompressed baseline TIFF)."""
import torch

from ..ops import UemError, call, need_gpu, ptr, stream
from ..utils import tiff


def edge_shrinking(label_supixl, win_size=3, region_size=16):
    """label (H,W) or (B,H,W) integer tensor on the device -> same shape, int32: the id where the whole
    (2*win_size+1)^2 window agrees, else int(H/region_size * W/region_size)  (superpixels.py:129-150)."""
    need_gpu(label_supixl)
    lab = label_supixl
    squeeze = lab.dim() == 2
    if squeeze:
        lab = lab.unsqueeze(0)
    if lab.dim() != 3:
        raise UemError("edge_shrinking: label must be (H,W) or (B,H,W)")
    lab = lab.to(torch.int32).contiguous()
    B, H, W = lab.shape
    cnt_sup = int(H / region_size * W / region_size)
    out = torch.empty_like(lab)
    call("uem_superpixel_shrink", ptr(lab), ptr(out), B, H, W, int(win_size), cnt_sup, stream())
    return out[0] if squeeze else out


def save_superpixels(path, label):
    """Write a (H,W) id map as the reference does (int32 `.tif`)."""
    tiff.write_tiff(path, label.detach().to("cpu", torch.int32).numpy())


def load_superpixels(path, device="cuda"):
    """Read `<name>.tif` -> (1,H,W) int64 tensor, the `label_t_sup` layout of the data loader (basedata.py:77-79,87)."""
    arr = tiff.read_tiff(path)
    return torch.from_numpy(arr.astype("int64")).unsqueeze(0).to(device)
