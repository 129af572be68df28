"""The solver's per-step tail on the HIP path.

The reference's training loop (solver.py:91-94) is
    loss = self.criterion(logits, a); self.optimizer.zero_grad(); loss.backward(); self.optimizer.step()
with `criterion` = nn.KLDivLoss() for mhb/mhb_coAtt and nn.CrossEntropyLoss() otherwise
(solver.py:25-28) and `optimizer` = torch.optim.Adam(model.parameters(), lr=cfg.lr) (solver.py:29).
The classes below keep those names, constructor defaults and call signatures, so the solver only
has to import them from here; their arithmetic runs in libvqa_fusion.so (csrc/train.hip).
"""
import torch
from torch import nn

from . import ops


def _times(d, g):
    """d (N,A) * g, g = the scalar gradient arriving at the loss (loss.backward(): a one-element GPU tensor)"""
    if g.is_cuda and g.dtype == torch.float32 and g.numel() == 1 and d.dim() == 2 and d.is_contiguous():
        return ops.scale_by_device_scalar(d, g.reshape(1))
    return d * g


class _CeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, d = ops.ce_loss(logits.contiguous(), target, want_grad=logits.requires_grad)
        ctx.save_for_backward(d)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _times(d, g), None


class _KlDivLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target):
        loss, d = ops.kldiv_loss(logp.contiguous(), target.contiguous(), want_grad=logp.requires_grad)
        ctx.save_for_backward(d)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _times(d, g), None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() as constructed at solver.py:28: mean over rows, ignore_index -100, no
    class weights / label smoothing.  forward(logits (N,A) fp32, target (N,) int64) -> scalar."""

    def forward(self, logits, target):
        return _CeLossFn.apply(logits, target)


class KLDivLoss(nn.Module):
    """nn.KLDivLoss() as constructed at solver.py:26: default reduction, i.e. the mean over all N*A
    elements of target * (log target - input).  forward(log-probs (N,A), target (N,A)) -> scalar."""

    def forward(self, logp, target):
        return _KlDivLossFn.apply(logp, target)


def criterion_for(model_name):
    """solver.py:25-28."""
    return KLDivLoss() if model_name in ("mhb_coAtt", "mhb") else CrossEntropyLoss()


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) (solver.py:29)
    with every parameter of a group updated by one vqf_adam_step call.  State layout
    ('step', 'exp_avg', 'exp_avg_sq') and param_groups match torch's, so `param_group['lr'] = ...`
    (solver.py:47-50) and optimizer state_dicts interchange with torch.optim.Adam."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if eps < 0.0:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {}".format(betas))
        if weight_decay < 0.0:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(int(st["step"].item()), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            beta1, beta2 = group["betas"]
            for step, items in by_step.items():
                ops.adam_step([i[0] for i in items], [i[1] for i in items], [i[2] for i in items],
                              [i[3] for i in items], step, group["lr"], beta1, beta2, group["eps"],
                              group["weight_decay"])
        return loss
