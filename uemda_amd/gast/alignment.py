"""Aligner (prototypes + online pseudo-label refinement) and DownscaleLabel on the MI355X.

Drop-in for the parts of reference `uemda/gast/alignment.py` that `tools/train_ssl_uem.py` uses:
`Aligner.__init__` (:26-77), `label_refine` (:194-293, modes all / s / p / l), `update_prototype`
(:86-90, 328-355), `update_avg` / `init_avg` (:107-126), `_pearson_dist` (:424-451), `DownscaleLabel`
(:484-509).  Stage-2-only losses (CORAL, whitening, class / instance alignment) are out of scope.
"""
import torch
import torch.nn as nn

from .. import _lib, ops
from ..ops import UemError, call, ptr, stream
from ..scatter import index_max
from . import pseudo_generation

_MODES = {"all": 0, "s": 1, "p": 2, "l": 3}
_NO_SUP_CHECK = __import__("os").environ.get("UEM_NO_SUP_CHECK", "0") != "0"     # diagnostic only


class DownscaleLabel(nn.Module):
    def __init__(self, scale_factor=16, n_classes=7, ignore_label=-1, min_ratio=0.75):
        super().__init__()
        assert scale_factor > 1
        self.scale_factor, self.n_classes = scale_factor, n_classes
        self.ignore_label, self.min_ratio = ignore_label, min_ratio

    def forward(self, label):
        ops.need_gpu(label)
        if label.dim() == 4:
            label = label.squeeze(dim=1)
        assert label.dim() == 3
        label = label.contiguous().long()
        b, H, W = label.shape
        s = self.scale_factor
        out = torch.empty((b, 1, H // s, W // s), device=label.device, dtype=torch.int64)
        call("uem_downscale_label", ptr(label), ptr(out), b, H, W, s, self.n_classes, int(self.ignore_label),
             float(self.min_ratio), stream())
        return out


class Aligner:
    def __init__(self, logger=None, feat_channels=64, class_num=7, ignore_label=-1, decay=0.999, topk=32,
                 resume=None, device="cuda", process_group=None):
        self.feat_channels, self.class_num, self.ignore_label = feat_channels, class_num, ignore_label
        self.decay, self.logger, self.eps, self.topk = decay, logger, 1e-7, topk
        self.device = torch.device(device)
        self.process_group = process_group
        if resume:
            self.prototypes = torch.load(resume, map_location='cpu').to(self.device).float().contiguous()
            if logger is not None:
                logger.info('finish init prototypes!')
        else:
            self.prototypes = torch.zeros([class_num, feat_channels], device=self.device)
        self.downscale_gt = DownscaleLabel(scale_factor=16, n_classes=class_num, ignore_label=ignore_label,
                                           min_ratio=0.75)
        self._data_sum = torch.zeros([class_num, feat_channels], device=self.device)
        self._data_cnt = torch.zeros([class_num, 1], device=self.device)

    # ---- domain alignment (stage 1/2, --align-domain) -------------------------------------------------------
    def align_domain(self, feat_s, feat_t):
        """CORAL loss between source and target feature maps (alignment.py:79-84)."""
        from ..loss import ops_as_rows
        from .coral import CoralLoss
        assert feat_s.shape == feat_t.shape, 'tensor "feat_s" has the same shape as tensor "feat_t"'
        assert len(feat_s.shape) == 4, 'tensor "feat_s" and "feat_t" must have 4 dimensions'
        if not hasattr(self, "coral"):
            self.coral = CoralLoss()
        return self.coral(ops_as_rows(feat_s, self.feat_channels), ops_as_rows(feat_t, self.feat_channels))

    # ---- distances ------------------------------------------------------------------------------------
    def _pearson_dist(self, feat1, feat2):
        """(n, k) x (m, k) -> (n, m) Pearson distance in [0, 1]  (alignment.py:424-451)."""
        ops.need_gpu(feat1, feat2)
        a, b = feat1.detach().contiguous().float(), feat2.detach().contiguous().float()
        n, k = a.shape
        m = b.shape[0]
        out = torch.empty((n, m), device=a.device, dtype=torch.float32)
        ws = torch.empty(m * k + m, device=a.device, dtype=torch.float32)
        call("uem_pearson_dist", ptr(a), ptr(b), ptr(out), ptr(ws), n, m, k, stream())
        return out

    def _pearson_sim_map(self, feat_nhwc):
        n, h, w, k = feat_nhwc.shape
        C = self.class_num
        sim = torch.empty((n, h, w, C), device=feat_nhwc.device, dtype=torch.float32)
        ws = torch.empty(C * k + C, device=feat_nhwc.device, dtype=torch.float32)
        call("uem_pearson_sim", ptr(feat_nhwc), ptr(self.prototypes), ptr(sim), ptr(ws), n * h * w, k, C, stream())
        return sim

    # ---- label refinement ------------------------------------------------------------------------------
    def label_refine(self, label_t_sup, feat_t, preds_t, label_t_soft, refine=True, mode='all', temp=2.0,
                     sup_ignore_id=None, return_plane_max=False, _select=None):
        """Three-view refinement of the soft pseudo label (alignment.py:194-293).

        `sup_ignore_id`: the ignored superpixel id; None reproduces the reference's batch-global
        `label_t_sup.max()` (computed on the device, no host sync).  Under data parallel pass the id the dataset's
        edge shrinking wrote -- cnt_sup = (h/16)*(w/16) of the FULL image the superpixel map was computed on
        (gast/superpixels.py:131,149), e.g. 1024 for 512x512 ISPRS tiles, 4096 for 1024x1024 LoveDA tiles cropped to
        512x512 -- so that ranks whose crops miss the ignored id still agree."""
        if mode == 'n':
            raise UemError("label_refine(mode='n'): the kNN view is an n x n cdist over n = B*h*w pixels, "
                           "infeasible at the benchmark batch and flagged unusable by the reference author")
        assert mode in _MODES
        if not refine:
            return label_t_soft
        ops.need_gpu(feat_t, label_t_soft)
        soft = label_t_soft.detach().contiguous().float()
        B, C, H, W = soft.shape
        feat = ops.as_nhwc(feat_t.detach())
        _, h, w, k = feat.shape
        dev = soft.device
        sim = lg1 = lg2 = sup = seg = ign = None
        S = 1
        if mode in ('all', 's'):
            sup = label_t_sup.detach().contiguous().long()
            self.check_superpixel_ids()                                # the previous call's range report, if any
            if sup_ignore_id is None:
                ign = index_max(sup)                                   # alignment.py:241 (device scalar, no host sync)
            else:
                ign = self._ignore_id_tensor(int(sup_ignore_id), dev)
            S = self._sup_table_size(H, W, sup_ignore_id)
            zeros = torch.zeros(B * S * C + 4, device=dev, dtype=torch.int32)      # the segment table and the range flag: one fill
            seg, oor = zeros[:B * S * C].view(B, S, C), zeros[B * S * C]
            # (the segment pass does not depend on the Pearson map below; run beside it on a second stream the pair took 17 us MORE
            # than back to back -- 311 against 294 us for the whole call at B = 32: the fork and the join cost more than the overlap of
            # two 55 us HBM-bound kernels gives)
            call("uem_segment_max_planar", ptr(soft), ptr(sup), ptr(seg), B, C, H, W, S, ptr(oor), stream())
        if mode in ('all', 'p'):
            sim = self._pearson_sim_map(feat)
        if mode in ('all', 'l'):
            if isinstance(preds_t, (list, tuple)):
                assert len(preds_t) == 2
                lg1, lg2 = ops.as_nhwc(preds_t[0].detach()).contiguous(), ops.as_nhwc(preds_t[1].detach()).contiguous()
            else:
                lg1 = ops.as_nhwc(preds_t.detach()).contiguous()
        if seg is not None:
            self._report_superpixel_range(oor, S)
        out = torch.empty_like(soft)
        plane_max = torch.empty((B, C), device=dev, dtype=torch.int32)
        ws = torch.empty(_lib.load().uem_label_refine_workspace_floats(B, C, H, W, S if seg is not None else 0), device=dev, dtype=torch.float32)
        self._last_plane_max = plane_max
        if _select is not None and (H * W) % 4 == 0:
            # label_refine + pseudo_selection in three launches (uem_label_refine_select): same numbers, the selection reads 5 bytes
            # per pixel instead of the refined map
            top, low, ignore = _select
            cand = torch.empty(_lib.load().uem_label_refine_select_workspace_bytes(B, H, W), device=dev, dtype=torch.uint8)
            hard = torch.empty((B, H, W), device=dev, dtype=torch.int64)
            call("uem_label_refine_select", ptr(soft), ptr(sup), ptr(sim), ptr(lg1), ptr(lg2), ptr(seg), ptr(ign), ptr(out),
                 ptr(plane_max), ptr(ws), ptr(cand), ptr(hard), B, C, h, w, H, W, S, float(temp), _MODES[mode], float(top), float(low),
                 int(ignore), stream())
            return out, hard
        call("uem_label_refine", ptr(soft), ptr(sup), ptr(sim), ptr(lg1), ptr(lg2), ptr(seg), ptr(ign), ptr(out),
             ptr(plane_max), ptr(ws), B, C, h, w, H, W, S, float(temp), _MODES[mode], stream())
        if _select is not None:
            top, low, ignore = _select
            return out, pseudo_generation.pseudo_selection(out, top, low, 'tensor', ignore, _plane_max=plane_max, check_range=False)
        return (out, plane_max) if return_plane_max else out

    def _ignore_id_tensor(self, value, dev):
        """the ignored superpixel id as the device scalar the kernels read, made once per (value, device)"""
        cache = self.__dict__.setdefault("_ign_cache", {})
        key = (value, dev.index)
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.full((), value, device=dev, dtype=torch.int64)
        return t

    # ---- superpixel table capacity ------------------------------------------------------------------------
    # The reference sizes the scatter from every batch's own maximum id (alignment.py:241-245, one host sync per
    # call).  Here the table has a capacity fixed WITHOUT looking at the data -- ids of a 1024x1024 LSC map (region
    # 16: up to 4096 segments + the ignored id 4096, gast/superpixels.py:131,149) or 4x the crop's own grid,
    # whichever is larger, `sup_capacity` if the dataset's maps are denser -- and the kernels report ids beyond it
    # through a device flag that is copied back asynchronously and checked at the next call / check_superpixel_ids():
    # a too-small table raises UemError (pixels with such ids were left unrefined, never given another segment's
    # maxima) instead of silently reading the wrong segment.
    sup_capacity = 0

    def _sup_table_size(self, H, W, sup_ignore_id):
        S = max(4 * (H // 16) * (W // 16) + 1, 4097, int(self.sup_capacity))
        if sup_ignore_id is not None:
            S = max(S, int(sup_ignore_id) + 1)
        return S

    def _report_superpixel_range(self, oor, S):
        if torch.cuda.is_current_stream_capturing():
            # under hipGraph capture no event of ours may be recorded and nothing may be read on the host: the flag stays on the
            # device (`last_superpixel_range_flag`) for the caller to look at after a replay
            self.last_superpixel_range_flag, self._oor_pending = oor, None
            return
        host = torch.empty((), dtype=torch.int32, pin_memory=True)
        host.copy_(oor, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._oor_pending = (ev, host, S, oor)

    def check_superpixel_ids(self, wait=True):
        """Raise if the last label_refine saw a superpixel id outside its table.  wait=True waits for that call (only); wait=False
        looks at the report only if it has already arrived and otherwise leaves it pending -- the next label_refine (or a later
        call here) picks it up, one step late and without stalling the host in the middle of a step (the eager step's form:
        VERDICT r4, the host's enqueue ran in lock step with the device)."""
        pend = getattr(self, "_oor_pending", None)
        if pend is None or _NO_SUP_CHECK:
            return
        if torch.cuda.is_current_stream_capturing():
            return                           # no event may be queried or waited for while a hipGraph is being captured
        ev, host, S, _keep = pend
        if not wait and not ev.query():
            return
        self._oor_pending = None
        ev.synchronize()
        worst = int(host)
        if worst != 0:
            raise UemError(f"label_refine: superpixel id {'< 0 or >= 2^31' if worst == 0x7fffffff else worst} is outside the "
                           f"segment table of {S} entries; the pixels carrying such ids were left unrefined. Set "
                           f"aligner.sup_capacity to the dataset's superpixel count + 1 (cnt_sup of the full image, "
                           f"reference gast/superpixels.py:131) and rerun the step")

    def refine_and_select(self, label_t_sup, feat_t, preds_t, label_t_soft, mode='all', temp=2.0, cutoff_top=0.8,
                          cutoff_low=0.6, sup_ignore_id=None):
        """label_refine + pseudo_selection sharing the per-class maxima computed inside the fused kernel
        (saves one full pass over the (B,C,H,W) map); identical results to calling the two in sequence."""
        return self.label_refine(label_t_sup, feat_t, preds_t, label_t_soft, True, mode, temp, sup_ignore_id,
                                 _select=(cutoff_top, cutoff_low, self.ignore_label))

    # ---- prototypes --------------------------------------------------------------------------------------
    def _class_sums(self, feat, label_ds):
        feat = ops.as_nhwc(feat.detach())
        n, h, w, k = feat.shape
        C = self.class_num
        lab = label_ds.contiguous().view(-1)
        sums = torch.empty((C, k), device=feat.device, dtype=torch.float32)
        cnts = torch.empty((C,), device=feat.device, dtype=torch.float32)
        ws = torch.empty(_lib.UEM_PROTO_SPLIT * C * (k + 1), device=feat.device, dtype=torch.float32)
        call("uem_proto_sums", ptr(feat), ptr(lab), ptr(sums), ptr(cnts), ptr(ws), n * h * w, k, C,
             int(self.ignore_label), stream())
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size(self.process_group) > 1:
            # replicas must see the same prototypes: reduce the partial sums BEFORE the division / EMA
            packed = torch.cat([sums.view(-1), cnts])
            torch.distributed.all_reduce(packed, group=self.process_group)
            sums, cnts = packed[:C * k].view(C, k), packed[C * k:]
        return sums, cnts

    def update_prototype(self, feat, label):
        """EMA update from source features + labels; returns the downscaled label (alignment.py:86-90)."""
        label = self.downscale_gt(label)
        sums, cnts = self._class_sums(feat, label)
        k = sums.shape[1]
        self.prototypes = self.prototypes.contiguous()
        call("uem_proto_ema", ptr(sums), ptr(cnts), ptr(self.prototypes), k, self.class_num, float(self.decay), stream())
        return label

    def update_avg(self, feat, label):
        label = self.downscale_gt(label)                                       # alignment.py:107-119
        sums, cnts = self._class_sums(feat, label)
        ops.add_(self._data_sum, sums)
        ops.add_(self._data_cnt, cnts.view(-1, 1).contiguous())

    def init_avg(self):
        """prototypes = accumulated class sums / (counts + eps)  (alignment.py:121-122; tools/init_prototypes.py)."""
        self.prototypes = torch.empty_like(self._data_sum)
        call("uem_proto_mean", ptr(self._data_sum), ptr(self._data_cnt.contiguous()), ptr(self.prototypes),
             self.feat_channels, self.class_num, stream())
        if self.logger is not None:
            self.logger.info('finish init prototypes!')
