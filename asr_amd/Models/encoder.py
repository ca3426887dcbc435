"""BLSTM encoder on the HIP kernels -- mirror of Models/encoder.py:14-35.

Same constructor arguments (read from ``hp``), same parameter names
(``bi_lstm.weight_ih_l0`` ... ``bi_lstm.bias_hh_l4_reverse``), same
``forward(x, lengths) -> hbatch (B,T,2H)`` contract: every sequence runs over
its own ``lengths[b]`` frames in both directions, rows past the length are 0.
No packing: the recurrence kernels mask by length (asr_blstm_fwd/bwd), the
input projections of all frames are MFMA GEMMs (asr_sgemm).
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib, ops, streams
from .._lib import check


class BiLSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM's names, shapes and default init
    (U(-1/sqrt(H), 1/sqrt(H)), drawn in nn.LSTM's registration order so the same
    seed gives the same weights as the reference's nn.LSTM)."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        H = hidden_size
        for layer in range(num_layers):
            d_in = input_size if layer == 0 else 2 * H
            for sfx in ("", "_reverse"):
                self.register_parameter(f"weight_ih_l{layer}{sfx}", nn.Parameter(torch.empty(4 * H, d_in)))
                self.register_parameter(f"weight_hh_l{layer}{sfx}", nn.Parameter(torch.empty(4 * H, H)))
                self.register_parameter(f"bias_ih_l{layer}{sfx}", nn.Parameter(torch.empty(4 * H)))
                self.register_parameter(f"bias_hh_l{layer}{sfx}", nn.Parameter(torch.empty(4 * H)))
        stdv = 1.0 / math.sqrt(H) if H > 0 else 0
        for p in self.parameters():
            nn.init.uniform_(p, -stdv, stdv)

    def flat(self):
        out = []
        for layer in range(self.num_layers):
            for sfx in ("", "_reverse"):
                for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    out.append(getattr(self, f"{kind}_l{layer}{sfx}"))
        return out


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Workspaces of the persistent recurrence kernels (flag words + h exchange), one
# per (device, B, H), reused by every layer and step: launches on one stream are
# serialised, and the library zeroes the flag words before each launch.  Word
# ERR_WORD is sticky: a kernel sets it if an in-launch wait ever timed out.
PERSISTENT = True          # False selects the one-launch-per-time-step kernels
ERR_WORD = 1024
_WS = {}


def _persist_ws(lib, B, H, dev):
    if not PERSISTENT:
        return None
    key = (dev.index, B, H)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev, dtype=torch.float32)
        _WS[key] = ws
    return ws


def persistent_error_flags():
    """True if any persistent kernel reported a timed-out wait (device -> host copy)."""
    return any(bool(ws.view(torch.int32)[ERR_WORD].item()) for ws in _WS.values())


class _BLSTMLayerFunction(torch.autograd.Function):
    """One bidirectional layer.  Input is (B,T,D) batch-major for the first layer
    and (T,B,D) time-major otherwise; output is (B,T,2H) batch-major for the last
    layer and (T,B,2H) time-major otherwise (one time step = one contiguous slab
    for the recurrence kernels).  One autograd node per layer, so a layer's
    gradients are final -- and its all-reduce bucket can start -- while the layers
    below are still in their backward."""

    @staticmethod
    def forward(ctx, inp, lengths_dev, first, last, H, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r,
                b_ih_r, b_hh_r):
        lib = _lib.load()
        inp = inp.contiguous()
        if first:
            B, T, D = inp.shape
            in_two = (B, T * D, D)         # row (t*B + b) of the GEMM <-> inp[b, t]
        else:
            T, B, D = inp.shape
            in_two = (0, D, 0)
        dev = inp.device
        st = _stream()
        G = torch.empty(T, B, 2, 4 * H, device=dev, dtype=torch.float32)
        Cs = torch.empty(T, B, 2, H, device=dev, dtype=torch.float32)
        for d, (w_ih, b_ih, b_hh) in enumerate(((w_ih_f, b_ih_f, b_hh_f), (w_ih_r, b_ih_r, b_hh_r))):
            ops.sgemm(inp, w_ih, G, T * B, 4 * H, D, a_two=in_two, b_stride=D, c_stride=8 * H,
                      c_off=d * 4 * H, bias1=b_ih, bias2=b_hh)
        W_hh = torch.cat([w_hh_f, w_hh_r], 0)      # [2][4H][H]
        if last:
            Hout = torch.empty(B, T, 2 * H, device=dev, dtype=torch.float32)
            ho_t, ho_b = 2 * H, T * 2 * H
        else:
            Hout = torch.empty(T, B, 2 * H, device=dev, dtype=torch.float32)
            ho_t, ho_b = B * 2 * H, 2 * H
        ws = _persist_ws(lib, B, H, dev)
        check(lib.asr_blstm_fwd(W_hh.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), ho_t, ho_b,
                                lengths_dev.data_ptr(), T, B, H, 0 if ws is None else ws.data_ptr(), st))
        ctx.save_for_backward(inp, G, Cs, Hout, lengths_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r)
        ctx.dims = (B, T, D, H, first, in_two, ho_t, ho_b)
        return Hout

    @staticmethod
    def backward(ctx, dHout):
        lib = _lib.load()
        B, T, D, H, first, in_two, ho_t, ho_b = ctx.dims
        inp, G, Cs, Hout, lengths_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r = ctx.saved_tensors
        dev = inp.device
        st = _stream()
        dHout = dHout.contiguous()                 # same layout as Hout
        f32 = dict(device=dev, dtype=torch.float32)
        dc_ws = torch.empty(2 * B * H, **f32)
        W_hh = torch.cat([w_hh_f, w_hh_r], 0).view(2, 4 * H, H)
        W_hhT = ops.transpose(W_hh)                # [2][H][4H] (step-wise kernels)
        ws = _persist_ws(lib, B, H, dev)
        check(lib.asr_blstm_bwd(W_hh.data_ptr(), W_hhT.data_ptr(), G.data_ptr(), Cs.data_ptr(), dHout.data_ptr(), ho_t, ho_b,
                                lengths_dev.data_ptr(), dc_ws.data_ptr(), T, B, H,
                                0 if ws is None else ws.data_ptr(), st))
        dG = G.view(T * B, 8 * H)                  # now d loss / d pre-activation gates (0 at padding)
        # critical path first: the input gradient feeds the next layer's recurrence
        dIn = None
        if ctx.needs_input_grad[0]:
            # dX = dG[:, fwd] W_ih_f + dG[:, rev] W_ih_r, written in the input's own layout
            dIn = torch.empty_like(inp)
            for d, w_ih in enumerate((w_ih_f, w_ih_r)):
                ops.sgemm(dG, w_ih, dIn, T * B, D, 4 * H, a_stride=8 * H, a_off=d * 4 * H, b_mode=1,
                          b_stride=D, c_two=in_two, accumulate=(d == 1))
        # weight / bias gradients: nothing downstream reads them -> side stream, overlapping the
        # next layer's (latency-bound) recurrence kernel
        hout_two = (B, ho_b, ho_t)                 # row (t*B + b) <-> Hout element offset
        grads = {}
        with streams.on_side(dev, G, inp, Hout):
            db = torch.empty(8 * H, **f32)
            ops.colsum(dG, db)                     # b_ih and b_hh receive the same column sums
            for d in range(2):
                # dW_hh[d] = sum_t dG_t^T h_{t-1}: forward pairs (t, t-1), reverse pairs (t, t+1);
                # padded rows of dG and Hout are zero, so ragged lengths need no masking here
                gW = torch.empty(4 * H, H, **f32)
                if T > 1:
                    K = (T - 1) * B
                    sk = ops.pick_splitk(4 * H, H, K)
                    if sk > 1:
                        gW.zero_()
                    a_off = d * 4 * H + (B * 8 * H if d == 0 else 0)
                    b_off = d * H + (0 if d == 0 else ho_t)
                    ops.sgemm(dG, Hout, gW, 4 * H, H, K, a_mode=1, a_stride=8 * H, a_off=a_off, b_mode=1,
                              b_two=hout_two, b_off=b_off, c_stride=H, splitk=sk)
                else:
                    gW.zero_()
                # dW_ih[d] = dG[:, d]^T X
                gWi = torch.empty(4 * H, D, **f32)
                K = T * B
                sk = ops.pick_splitk(4 * H, D, K)
                if sk > 1:
                    gWi.zero_()
                ops.sgemm(dG, inp, gWi, 4 * H, D, K, a_mode=1, a_stride=8 * H, a_off=d * 4 * H, b_mode=1,
                          b_two=in_two, c_stride=D, splitk=sk)
                grads[d] = (gWi, gW, db[d * 4 * H:(d + 1) * 4 * H], db[d * 4 * H:(d + 1) * 4 * H].clone())
        return (dIn, None, None, None, None) + grads[0] + grads[1]


class _DropoutFunction(torch.autograd.Function):
    """Inverted dropout between layers (nn.LSTM(dropout=p), Models/encoder.py:26)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.numel(), device=x.device, dtype=torch.uint8)
        check(lib.asr_dropout_fwd(x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), C.c_float(p),
                                  C.c_uint64(seed), _stream()))
        ctx.mask, ctx.p = mask, p
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.asr_dropout_bwd(dy.data_ptr(), ctx.mask.data_ptr(), dx.data_ptr(), dy.numel(),
                                  C.c_float(ctx.p), _stream()))
        return dx, None, None


class Encoder(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.num_layers = hp.num_encoder_layer
        self.hidden = hp.num_hidden_nodes_encoder
        self.dropout = float(hp.encoder_dropout)
        self.bi_lstm = BiLSTMParams(hp.lmfb_dim * hp.frame_stacking, self.hidden, self.num_layers)
        self._step = 0

    def forward(self, x, lengths):
        if not x.is_cuda:
            raise RuntimeError("asr_amd.Encoder: input must be on the GPU; the HIP path has no CPU fallback")
        lengths_dev = lengths.to(device=x.device, dtype=torch.int32)
        p = self.dropout if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0
        h = x.float()
        flat = self.bi_lstm.flat()
        for layer in range(self.num_layers):
            last = layer == self.num_layers - 1
            h = _BLSTMLayerFunction.apply(h, lengths_dev, layer == 0, last, self.hidden,
                                          *flat[8 * layer: 8 * layer + 8])
            if p > 0.0 and not last:
                h = _DropoutFunction.apply(h, p, seed + 7919 * layer)
        return h
