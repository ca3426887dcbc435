"""Fused gradient clipping + Adam step on the HIP kernels -- the caller-side tail of
the reference's training step (train.py:60-63, optimizer built at train.py:146):

    torch.nn.utils.clip_grad_norm_(model.parameters(), hp.clip)
    optimizer.step()            # torch.optim.Adam(lr=hp.lr, weight_decay=1e-5)

``ClipAdam`` is a ``torch.optim.Optimizer`` with torch.optim.Adam's constructor
arguments, state layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter) and
``state_dict`` format, so optimizer checkpoints interchange with the reference's
(train.py:93-101 saves and re-creates the optimizer).  ``step(max_norm=...)`` clips the
global L2 norm over ALL parameter groups first (what clip_grad_norm_ over
model.parameters() does), then updates: three launches per 48 tensors, no host
synchronisation (the clip coefficient never leaves the device).  Like every op of
``asr_amd`` it needs the GPU library: parameters on the CPU raise.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import TENSOR_LIST_MAX, TensorList, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tensor_list(tensors):
    tl = TensorList()
    tl.count = len(tensors)
    for i, t in enumerate(tensors):
        tl.p[i] = t.data_ptr()
        tl.n[i] = t.numel()
    return tl


def _check_tensor(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"asr_amd.ClipAdam: {what} must live on the GPU (no CPU path exists)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"asr_amd.ClipAdam: {what} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"asr_amd.ClipAdam: {what} must be contiguous")


def grad_sqnorm(grads, out=None):
    """Device scalar: sum of squares of all `grads` (deterministic)."""
    lib = _lib.load()
    dev = grads[0].device
    if out is None:
        out = torch.empty(1, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.asr_sqnorm_ws_floats()), device=dev, dtype=torch.float32)
    for k in range(0, len(grads), TENSOR_LIST_MAX):
        tl = _tensor_list(grads[k:k + TENSOR_LIST_MAX])
        check(lib.asr_grad_sqnorm(C.byref(tl), out.data_ptr(), int(k > 0), ws.data_ptr(), _stream()))
    return out


class ClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("ClipAdam: lr, eps and weight_decay must be non-negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"ClipAdam: invalid betas {betas}")
        # the keys torch.optim.Adam keeps in a param_group, so state_dicts interchange
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self.max_norm = max_norm

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """Clip (if max_norm, or the constructor's max_norm, is set) and update.  Returns the
        total gradient norm before clipping as a 0-dim device tensor (what clip_grad_norm_
        returns), or None without clipping."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        max_norm = self.max_norm if max_norm is None else max_norm
        groups = []
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("ClipAdam: amsgrad / maximize are not on the reference's path")
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                _check_tensor(p, "a parameter")
                _check_tensor(p.grad, "a gradient")
            groups.append((group, ps))
        every = [p for _, ps in groups for p in ps]
        if not every:
            return loss
        norm2 = None
        if max_norm is not None and max_norm > 0:
            norm2 = grad_sqnorm([p.grad for p in every])
        for group, ps in groups:
            if not ps:
                continue
            steps = set()
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)      # torch.optim.Adam's layout
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(int(st["step"]))
                _check_tensor(st["exp_avg"], "exp_avg")
                _check_tensor(st["exp_avg_sq"], "exp_avg_sq")
            b1, b2 = group["betas"]
            for t in sorted(steps):       # normally one value; parameters added later may lag
                sel = [p for p in ps if int(self.state[p]["step"]) == t]
                for k in range(0, len(sel), TENSOR_LIST_MAX):
                    part = sel[k:k + TENSOR_LIST_MAX]
                    lists = [_tensor_list(part), _tensor_list([p.grad for p in part]),
                             _tensor_list([self.state[p]["exp_avg"] for p in part]),
                             _tensor_list([self.state[p]["exp_avg_sq"] for p in part])]
                    check(lib.asr_clip_adam_step(C.byref(lists[0]), C.byref(lists[1]), C.byref(lists[2]),
                                                 C.byref(lists[3]), 0 if norm2 is None else norm2.data_ptr(),
                                                 C.c_double(max_norm if norm2 is not None else 0.0),
                                                 C.c_double(group["lr"]), C.c_double(b1), C.c_double(b2),
                                                 C.c_double(group["eps"]), C.c_double(group["weight_decay"]), t,
                                                 _stream()))
        self.last_total_norm = None if norm2 is None else norm2.sqrt().squeeze(0)
        return loss if closure is not None else self.last_total_norm
