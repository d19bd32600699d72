"""Synthetic tiles of SURVEY.md section 8(d): seeded, generated on the CPU so that every implementation and both
machines see bit-identical inputs.  Used by bench.py, the scripts and (through `oracle/synth.py`) the tests."""
import torch
import torch.nn.functional as F


def make_batch(B=2, H=256, W=256, C=6, k=2048, seed=2333, scale=16):
    g = torch.Generator().manual_seed(seed)
    h, w = H // scale, W // scale
    images_s = torch.randn(B, 3, H, W, generator=g)
    # target pipeline clamps normalised pixels to <= 1.0 (reference aug/augmentation.py:112-122)
    images_t = torch.randn(B, 3, H, W, generator=g).clamp(max=1.0)
    # source labels: 16x16 blobs in {-1..C-1} with per-cell label noise, so DownscaleLabel sees
    # cells above and below its 0.75 majority threshold
    cell = torch.randint(-1, C, (B, 1, h, w), generator=g)
    label_s = F.interpolate(cell.float(), scale_factor=scale, mode="nearest").long().squeeze(1)
    noise_p = torch.rand(B, 1, h, w, generator=g) * 0.4
    noise_p = F.interpolate(noise_p, scale_factor=scale, mode="nearest").squeeze(1)
    flip = torch.rand(B, H, W, generator=g) < noise_p
    rnd = torch.randint(-1, C, (B, H, W), generator=g)
    label_s = torch.where(flip, rnd, label_s)
    # soft pseudo labels: peaked softmax at feature resolution, upsampled, plus pixel noise
    low = torch.softmax(3.0 * torch.randn(B, C, h, w, generator=g), dim=1)
    soft = F.interpolate(low, (H, W), mode="bilinear", align_corners=True)
    soft = soft * (1.0 + 0.05 * torch.rand(B, C, H, W, generator=g))
    label_t_soft = (soft / soft.sum(dim=1, keepdim=True)).contiguous()
    # superpixels: h*w blocks of 16x16 with a 3-px shrunk border set to the ignored id h*w
    # (reference gast/superpixels.py:129-152: 7x7 window => win 3, ignored id = H/16*W/16)
    yy = torch.arange(H).view(H, 1)
    xx = torch.arange(W).view(1, W)
    ids = (yy // scale) * w + (xx // scale)
    border = ((yy % scale) < 3) | ((yy % scale) >= scale - 3) | ((xx % scale) < 3) | ((xx % scale) >= scale - 3)
    sup = torch.where(border, torch.full_like(ids, h * w), ids)
    label_t_sup = sup.view(1, 1, H, W).expand(B, 1, H, W).contiguous()
    prototypes = torch.randn(C, k, generator=g)
    return dict(images_s=images_s, label_s=label_s, images_t=images_t, label_t_soft=label_t_soft,
                label_t_sup=label_t_sup, prototypes=prototypes)


def irregular_superpixels(B, H, W, n_seg, seed=7):
    """Irregular (Voronoi-like) superpixel labels with the ignored id = n_seg on region borders."""
    g = torch.Generator().manual_seed(seed)
    cy = torch.rand(B, n_seg, generator=g) * H
    cx = torch.rand(B, n_seg, generator=g) * W
    yy = torch.arange(H).view(1, H, 1, 1).float()
    xx = torch.arange(W).view(1, 1, W, 1).float()
    d = (yy - cy.view(B, 1, 1, n_seg)) ** 2 + (xx - cx.view(B, 1, 1, n_seg)) ** 2
    lab = d.argmin(dim=-1)                                             # (B, H, W)
    edge = torch.zeros_like(lab, dtype=torch.bool)
    edge[:, 1:, :] |= lab[:, 1:, :] != lab[:, :-1, :]
    edge[:, :, 1:] |= lab[:, :, 1:] != lab[:, :, :-1]
    lab = torch.where(edge, torch.full_like(lab, n_seg), lab)
    return lab.unsqueeze(1)
