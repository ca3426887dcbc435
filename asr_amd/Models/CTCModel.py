"""CTC model on the HIP kernels -- mirror of Models/CTCModel.py:8-47.

``CTCModel(hp)``: the BLSTM encoder of asr_amd.Models.encoder plus one Linear
(2*hp.num_hidden_nodes -> hp.num_classes) whose products run on asr_sgemm.  Parameter
names as in the reference (``encoder.bi_lstm.*``, ``decoder.weight``, ``decoder.bias``), so
checkpoints interchange.  ``encoder_type == 'Wave'`` (WaveEncoder) raises: out of scope.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import check
from .encoder import Encoder


class _LinearFunction(torch.autograd.Function):
    """y = x W^T + b over the last axis, x (..., K) contiguous."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.view(-1, K)
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        ops.linear_nt(x2, weight, y.view(-1, N), bias1=bias)
        ctx.save_for_backward(x2, weight)
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.contiguous().view(-1, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
            ops.matmul_nn(dy2, weight, dx.view(-1, K))
        if ctx.needs_input_grad[1]:
            dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
            ops.matmul_tn(dy2, x2, dw)
        if ctx.needs_input_grad[2]:
            db = torch.empty(N, device=dy.device, dtype=torch.float32)
            ops.colsum(dy2, db)
        return dx, dw, db


class Linear(nn.Module):
    """nn.Linear's parameters, names and default init (kaiming_uniform(a=sqrt 5), bias
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)), drawn in the same order)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("asr_amd.Linear: input must be on the GPU; the HIP path has no CPU fallback")
        return _LinearFunction.apply(x.float(), self.weight, self.bias)


class CTCModel(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        if getattr(hp, "encoder_type", None) == "Wave":
            raise NotImplementedError("asr_amd.CTCModel: the WaveEncoder variant is outside the HIP path")
        self.encoder = Encoder(hp)
        self.decoder = Linear(hp.num_hidden_nodes * 2, hp.num_classes)

    def forward(self, x, lengths, targets=None, pos_mel=None):
        hbatch = self.encoder(x, lengths)
        return self.decoder(hbatch)          # (B, T, V); padded frames hold the bias, as in the reference

    def frame_labels(self, x, lengths):
        """(B,T) int64 argmax labels of every frame (device tensor)."""
        lib = _lib.load()
        with torch.no_grad():
            y = self.forward(x, lengths)
            B, T, V = y.shape
            out = torch.empty(B, T, device=y.device, dtype=torch.int64)
            check(lib.asr_argmax_rows(y.data_ptr(), B * T, V, V, out.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def decode(self, x, lengths, model_lm=None, src_pos=None):
        """Greedy CTC decoding, Models/CTCModel.py:25-42: collapse repeats, drop blanks (id 0).
        As there, every one of the T frames of a row is read (padded frames included) and
        ``prev_id`` carries over from one utterance to the next.  (The reference reads a global
        ``hp`` here that its import line comments out; ``self.hp`` stands in.)"""
        ids = self.frame_labels(x, lengths).cpu().tolist()      # one device -> host copy
        results = []
        prev_id = self.hp.num_classes + 1
        for row in ids:
            res = []
            for v in row:
                if v != prev_id and v != 0:
                    res.append(v)
                prev_id = v
            results.append(res)
        return results

    def align(self, x, lengths, targets=None, blank_id=0, space_id=1, sos_id=2, eos_id=1):
        with torch.no_grad():
            return self.forward(x, lengths)
