"""MaskedAdam -- the reference's grid optimizer (lib/masked_adam.py:18-71) over the gfx950 streaming kernels of
csrc/k4_opt.hip.  Same constructor, param-group keys (`lr`, `betas`, `eps`, `skip_zero_grad`), state keys (`step`,
`exp_avg`, `exp_avg_sq`), `set_pervoxel_lr` and kernel selection order as upstream; there is no CPU path."""
import torch

from .. import _native as N


def _launch(name, param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    for t in (param, grad, exp_avg, exp_avg_sq) + (() if perlr is None else (perlr,)):
        if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError(f'{name}: tensors must be contiguous fp32 device tensors')   # adam_upd.cpp CHECK_INPUT
        if t.numel() != param.numel():
            raise ValueError(f'{name}: size mismatch')
    fn = getattr(N.lib(), name)
    st = N.stream()
    if perlr is None:
        rc = fn(N.ptr(param), N.ptr(grad), N.ptr(exp_avg), N.ptr(exp_avg_sq), param.numel(), int(step),
                float(beta1), float(beta2), float(lr), float(eps), st)
    else:
        rc = fn(N.ptr(param), N.ptr(grad), N.ptr(exp_avg), N.ptr(exp_avg_sq), N.ptr(perlr), param.numel(), int(step),
                float(beta1), float(beta2), float(lr), float(eps), st)
    N.check(rc, name)


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    """adam_upd_cuda.adam_upd (lib/cuda/adam_upd.cpp:34-44)."""
    _launch('k4_adam_upd', param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps)


def masked_adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    """adam_upd_cuda.masked_adam_upd (lib/cuda/adam_upd.cpp:46-56)."""
    _launch('k4_masked_adam_upd', param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps)


def adam_upd_with_perlr(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    """adam_upd_cuda.adam_upd_with_perlr (lib/cuda/adam_upd.cpp:58-69)."""
    _launch('k4_adam_upd_with_perlr', param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps)


class MaskedAdam(torch.optim.Optimizer):
    """Adam with (1) per-voxel learning rate and (2) masked update that skips zero-gradient voxels."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        self.per_lr = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]['params'][0].shape == count.shape
        self.per_lr = (count.float() / count.max()).contiguous()

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            lr, (beta1, beta2), eps = group['lr'], group['betas'], group['eps']
            skip_zero_grad = group['skip_zero_grad']           # KeyError without it, as upstream (masked_adam.py:45)
            for param in group['params']:
                if param.grad is None:
                    continue
                state = self.state[param]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state['step'] += 1
                grad = param.grad if param.grad.is_contiguous() else param.grad.contiguous()
                if self.per_lr is not None and param.shape == self.per_lr.shape:
                    adam_upd_with_perlr(param, grad, state['exp_avg'], state['exp_avg_sq'], self.per_lr,
                                        state['step'], beta1, beta2, lr, eps)
                elif skip_zero_grad:
                    masked_adam_upd(param, grad, state['exp_avg'], state['exp_avg_sq'],
                                    state['step'], beta1, beta2, lr, eps)
                else:
                    adam_upd(param, grad, state['exp_avg'], state['exp_avg_sq'],
                             state['step'], beta1, beta2, lr, eps)
