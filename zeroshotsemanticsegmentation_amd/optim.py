"""optim.py -- fused Adam / SGD-momentum steps (one HIP launch per parameter tensor) with the interface and
state layout of torch.optim.Adam / torch.optim.SGD, so that the reference's optimizer wiring
(train.py:126-133,174-175: two parameter groups, bias lr x2, weight decay on weights only) and its
checkpoints (`optim_state_dict`, trainer_fcn.py:281-288) carry over unchanged.
"""
import torch

from . import _lib as L


def _dense_same_layout(p, g):
    """gradient with exactly the memory layout of p (the kernels walk raw storage)"""
    if g.dtype == torch.float32 and g.stride() == p.stride() and g.is_cuda:
        return g
    out = torch.empty_like(p)           # preserve_format keeps p's strides for dense tensors
    out.copy_(g)
    return out


def _state_like(p, t):
    """optimizer state with exactly p's memory layout.  Optimizer.load_state_dict keeps the strides of the checkpoint's
    tensors: a reference torch.optim checkpoint carries NCHW-contiguous moments while the parameters here are
    channels_last, and the kernels walk raw storage -- re-materialise such a tensor (logical copy) before use."""
    if t.stride() == p.stride() and t.dtype == torch.float32 and t.device == p.device:
        return t
    out = torch.empty_like(p)
    out.copy_(t)
    return out


def _check_dense(p):
    if not p.is_cuda:
        raise L.SznError("fused optimizers need GPU parameters (no CPU fallback)")
    if p.dtype != torch.float32:
        raise L.SznError("fused optimizers keep fp32 master parameters, got %s" % p.dtype)
    if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
        raise L.SznError("fused optimizers need dense parameters (contiguous or channels_last)")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad), update done by szn_adam_step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super(FusedAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        st = L.stream_ptr()
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                _check_dense(p)
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = torch.tensor(0.0)
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                for k in ('exp_avg', 'exp_avg_sq'):
                    state[k] = _state_like(p, state[k])
                step = int(state['step']) + 1
                state['step'] = torch.tensor(float(step))
                g = _dense_same_layout(p, p.grad)
                L.call("szn_adam_step", p.numel(), L.ptr(p), L.ptr(g), L.ptr(state['exp_avg']), L.ptr(state['exp_avg_sq']),
                       float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']), step,
                       float(grad_scale), None, 0, st)
                torch.autograd.graph.increment_version(p)      # weight images are refreshed lazily from this
        return loss


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD with momentum (dampening 0, no nesterov), update done by szn_sgd_momentum_step."""

    def __init__(self, params, lr, momentum=0, weight_decay=0):
        super(FusedSGD, self).__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        st = L.stream_ptr()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                _check_dense(p)
                state = self.state[p]
                first = 'momentum_buffer' not in state or state['momentum_buffer'] is None
                if first:
                    state['momentum_buffer'] = torch.zeros_like(p)
                else:
                    state['momentum_buffer'] = _state_like(p, state['momentum_buffer'])
                g = _dense_same_layout(p, p.grad)
                L.call("szn_sgd_momentum_step", p.numel(), L.ptr(p), L.ptr(g), L.ptr(state['momentum_buffer']),
                       float(group['lr']), float(group['momentum']), float(group['weight_decay']), int(first),
                       float(grad_scale), None, 0, st)
                torch.autograd.graph.increment_version(p)
        return loss
