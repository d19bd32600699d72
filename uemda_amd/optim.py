"""clip_grad_norm_ + SGD(momentum, weight_decay) fused over the model's flat arenas
(reference tools/train_ssl_uem.py:169-170,228-232: torch.optim.SGD + clip_grad.clip_grad_norm_(32))."""
import torch

from . import _lib, ops
from .ops import UemError, call, ptr, stream


def _arena_of(parameters):
    params = [p for p in parameters]
    owners = {id(getattr(p, "_uem_owner", None)) for p in params}
    owner = getattr(params[0], "_uem_owner", None) if params else None
    if owner is None or len(owners) != 1:
        raise UemError("parameters do not belong to one uemda_amd model arena (call model.cuda() first)")
    return owner


def grad_norm(model):
    """L2 norm of the flat gradient arena -> 1-element device tensor (two-stage deterministic reduce)."""
    arena, garena, n = model.flat_parameters()
    partial = torch.empty(_lib.UEM_NORM_BLOCKS, device=arena.device, dtype=torch.float32)
    norm = torch.empty(1, device=arena.device, dtype=torch.float32)
    call("uem_grad_sqnorm", ptr(garena), n, ptr(partial), ptr(norm), stream())
    return norm


def clip_grad_norm_(parameters, max_norm, norm_type=2):
    """Drop-in for torch.nn.utils.clip_grad_norm_ on a uemda_amd model: scales the gradient arena in
    place by min(max_norm / (norm + 1e-6), 1) and returns the total norm."""
    if norm_type != 2:
        raise UemError("clip_grad_norm_: only the L2 norm is implemented")
    model = _arena_of(parameters)
    arena, garena, n = model.flat_parameters()
    ops.grad_join()                # weight gradients still on the side / second stream (a backward that raised never ran its join callback)
    norm = grad_norm(model)
    coef = torch.clamp(float(max_norm) / (norm + 1e-6), max=1.0)      # one scalar
    call("uem_scale_by_scalar", ptr(garena), None, n, ptr(coef), stream())
    return norm[0]


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) semantics (no nesterov / dampening) in ONE pass over
    the flat parameter arena; `step(max_norm=...)` also fuses clip_grad_norm_ into that pass."""

    def __init__(self, model, lr=1e-2, momentum=0.9, weight_decay=5e-4):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        arena, _, _ = model.flat_parameters()
        self.momentum_buffer = torch.zeros_like(arena)
        self._steps = 0
        self.last_grad_norm = None

    def zero_grad(self, set_to_none=False):
        self.model.zero_grad()

    def _trainable_ranges(self, n):
        """[(offset, count)] runs of the flat arena that belong to trainable parameters: the whole arena unless something is frozen."""
        params = self.param_groups[0]["params"]
        if all(p.requires_grad for p in params):
            return [(0, n)]
        runs = []
        for p in params:
            if not p.requires_grad:
                continue
            off, cnt = p._uem_off, p.numel()
            if runs and runs[-1][0] + runs[-1][1] == off:
                runs[-1] = (runs[-1][0], runs[-1][1] + cnt)
            else:
                runs.append((off, cnt))
        return runs

    # learning rate as a device scalar (None: by value from param_groups): set by uemda_amd.step.GraphedStep, whose captured
    # optimizer launch would otherwise keep the learning rate of the capture
    lr_device = None

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, grad_prescale=1.0):
        g = self.param_groups[0]
        arena, garena, n = self.model.flat_parameters()
        ops.grad_join()            # no-op after a clean backward (its end-of-backward callback joined); the safety net after one that raised
        norm = None
        if max_norm is not None:
            norm = grad_norm(self.model)
            self.last_grad_norm = norm
        for off, cnt in self._trainable_ranges(n):
            # frozen parameters (requires_grad False: ResNetEncoder freeze_at / batchnorm_trainable) are skipped altogether, as
            # torch.optim.SGD skips a parameter whose .grad is None -- no weight decay, no momentum
            call("uem_sgd_clip_step", ptr(arena) + 4 * off, ptr(garena) + 4 * off, ptr(self.momentum_buffer) + 4 * off, cnt, ptr(norm),
                 float(max_norm) if max_norm is not None else 0.0, float(g["lr"]), float(g["momentum"]),
                 float(g["weight_decay"]), 1 if self._steps == 0 else 0, float(grad_prescale), ptr(self.lr_device), stream())
        self._steps += 1
        ops.weights_changed()
