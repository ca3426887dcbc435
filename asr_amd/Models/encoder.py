"""BLSTM encoder on the HIP kernels -- mirror of Models/encoder.py:14-35.

Same constructor arguments (read from ``hp``), same parameter names
(``bi_lstm.weight_ih_l0`` ... ``bi_lstm.bias_hh_l4_reverse``), same
``forward(x, lengths) -> hbatch (B,T,2H)`` contract: every sequence runs over
its own ``lengths[b]`` frames in both directions, rows past the length are 0.
No packed copies: the recurrence kernels mask by length (asr_blstm_fwd/bwd); the
input projections and their gradients are MFMA GEMMs (asr_sgemm) that run over a
row list of the frames inside their sequences, so padded frames cost no matrix
work -- what pack_padded_sequence (Models/encoder.py:32) buys the reference.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from .. import _lib, ops, persist, streams
from .._lib import check
from ..utils.utils import lengths_to_device


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


# Workspaces of the persistent recurrence kernels (flag words + h exchange): asr_amd/persist.py,
# one per (device, B, H, stream), reused by every layer and step on that stream; the library resets
# the flag words before each launch.  Word ERR_WORD is sticky: a kernel sets it if an in-launch
# wait ever timed out; label_smoothing_loss turns it into a NaN loss (persist.poison).
PERSISTENT = True          # False selects the one-launch-per-time-step kernels
ERR_WORD = persist.ERR_WORD
_WS = persist._WS


def _persist_ws(lib, B, H, dev):
    if not PERSISTENT:
        return None
    return persist.workspace(lib, B, H, dev)


persistent_error_flags = persist.error_flags


# Chunk pipelining (DESIGN.md "overlap"): a layer's recurrence can run as PIPE_CHUNKS launches
# over consecutive step ranges; after each launch the frames that are final -- the forward
# direction's first steps and the reverse direction's last frames -- are handed to a second
# stream, which in the forward pass applies the inter-layer dropout and accumulates their part of
# the NEXT layer's input projection, and in the backward pass accumulates their part of dX.
# Measured on MI355X (config 2): the chunked launches themselves cost +7 us per layer, but GEMM
# waves sharing the recurrence's CUs stretch every recurrence step by ~15% and the GEMMs by ~40%,
# which cancels what the overlap hides (39.3 ms/step off, 39.5-40.0 ms on) -- so the default is
# off (1); ASR_PIPE_CHUNKS=4 turns it on.
PIPE_CHUNKS = int(os.environ.get("ASR_PIPE_CHUNKS", "1"))
PIPE_MIN_T = 64
PIPE_FWD = os.environ.get("ASR_PIPE_FWD", "1") != "0"
PIPE_BWD = os.environ.get("ASR_PIPE_BWD", "1") != "0"


# Row lists: when at least this share of the (T x B) frames is padding, the layer GEMMs skip it.
# One launch for both directions of a layer's input projection and of its dX (two-level views
# over the separate forward / reverse parameter tensors); ASR_MERGE_DIRECTIONS=0 restores two.
MERGE_DIRECTIONS = os.environ.get("ASR_MERGE_DIRECTIONS", "1") != "0"
PACK_WEIGHTS = os.environ.get("ASR_PACK_WEIGHTS", "1") != "0"     # pre-split W_ih for the projection / dX products (A/B timing)
PACK_INPUT = os.environ.get("ASR_PACK_INPUT", "1") != "0"         # ... and the dropped-out layer input (projection, dW_ih)
ROW_LIST_MIN_PADDING = float(os.environ.get("ASR_ROW_LIST_MIN_PADDING", "0.04"))


class _RowLists:
    """Device lists (int32) of the frames t*B + b that lie inside their sequence, t-major:
    rows    -- all of them (n entries): input projections, dX, dW_ih;
    prev    -- (t-1)*B + b for the listed frames with t >= 1 (n_prev entries): the forward
               direction's dW_hh pairs dG[t] with h[t-1];
    rows[:n_next] -- the listed frames with t <= T-2: the reverse direction pairs dG[t], h[t+1]."""

    def __init__(self, rows, n, prev, n_prev, n_next):
        self.rows, self.n, self.prev, self.n_prev, self.n_next = rows, n, prev, n_prev, n_next


def _row_lists(lengths, T, B, dev):
    """None when the lengths live on the device already (building the lists would stall the host),
    or when the batch has too little padding for them to pay."""
    if lengths.is_cuda or T * B == 0:
        return None
    ln = lengths.detach().to(torch.int64).clamp(0, T)
    n = int(ln.sum())
    if n == 0 or n > (1.0 - ROW_LIST_MIN_PADDING) * T * B:
        return None
    mask = torch.arange(T).unsqueeze(1) < ln.unsqueeze(0)             # (T, B)
    rows = torch.nonzero(mask.flatten()).flatten().to(torch.int32)
    nb0, nbl = int(mask[0].sum()), int(mask[T - 1].sum())
    staged = torch.empty(2 * n - nb0, dtype=torch.int32, pin_memory=True)
    staged[:n] = rows
    staged[n:] = rows[nb0:] - B
    both = staged.to(dev, non_blocking=True)
    return _RowLists(both[:n], n, both[n:], n - nb0, n - nbl)


def _chunks(T, n):
    tc = -(-T // n)
    return [(s0, min(T, s0 + tc)) for s0 in range(0, T, tc)]


class _Next:
    """What a layer needs to prepare its successor's input: the successor's input weights and the
    dropout between the two.  Plain attribute bag, invisible to autograd on purpose: the successor
    computes the gradients of these weights itself from its saved input."""

    def __init__(self, w_ih_f, b_ih_f, b_hh_f, w_ih_r, b_ih_r, b_hh_r, p, seed):
        self.w = (w_ih_f.detach(), w_ih_r.detach())
        self.b = (b_ih_f.detach(), b_hh_f.detach(), b_ih_r.detach(), b_hh_r.detach())
        self.p, self.seed = p, seed
        self.G = None            # filled by the producing layer: (T,B,2,4H) pre-activations


class _BLSTMLayerFunction(torch.autograd.Function):
    """One bidirectional layer.  Input is (B,T,D) batch-major for the first layer
    and (T,B,D) time-major otherwise; output is (B,T,2H) batch-major for the last
    layer and (T,B,2H) time-major otherwise (one time step = one contiguous slab
    for the recurrence kernels).  One autograd node per layer, so a layer's
    gradients are final -- and its all-reduce bucket can start -- while the layers
    below are still in their backward.

    pre: _Next whose .G holds this layer's input projection (made by the layer below) or None;
    nxt: _Next to fill for the layer above (the output is then already dropped out) or None."""

    @staticmethod
    def forward(ctx, inp, lengths_dev, rl, first, last, H, pre, nxt, in_drop, whh_cat, w_ih_f, w_hh_f, b_ih_f, b_hh_f,
                w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        """in_drop = (p, seed) or None: inverted dropout applied to the input inside this node (the
        dropout between two nn.LSTM layers, Models/encoder.py:26), so that its backward fuses into
        the epilogue of the dX product."""
        lib = _lib.load()
        inp = inp.contiguous()
        in_keep, in_scale = None, 1.0
        inp_packed = False
        if in_drop is not None and in_drop[0] > 0.0:
            raw = inp
            in_scale = 1.0 / (1.0 - in_drop[0])
            # The dropped-out input feeds two matrix products and nothing else (this layer's input projection, and its
            # dW_ih in the backward): where asr_sgemm has pre-split kernels for both, the dropout writes {hi:lo} words
            # instead of fp32 -- the same bytes, and neither product splits the operand again (bit-identical results)
            Tn, Bn, Dn = raw.shape
            n_rows = rl.n if rl is not None else Tn * Bn
            k_dw = n_rows
            inp_packed = (PACK_INPUT and PACK_WEIGHTS and MERGE_DIRECTIONS and not first and pre is None and nxt is None
                          and ops.packed_ok(n_rows, 8 * H)
                          and ops.packed_dw_ok(8 * H, Dn, k_dw, ops.pick_splitk(8 * H, Dn, k_dw))
                          and None not in (ops.elem_delta(w_ih_f, w_ih_r), ops.elem_delta(b_ih_f, b_ih_r),
                                           ops.elem_delta(b_hh_f, b_hh_r)))
            if inp_packed:
                inp, in_keep = ops.dropout_fwd_packed(raw, in_drop[0], in_drop[1])
            else:
                inp = torch.empty_like(raw)
                in_keep = torch.empty(raw.numel(), device=raw.device, dtype=torch.uint8)
                check(lib.asr_dropout_fwd(raw.data_ptr(), inp.data_ptr(), in_keep.data_ptr(), raw.numel(),
                                          C.c_float(in_drop[0]), C.c_uint64(in_drop[1]), _stream()))
        if first:
            B, T, D = inp.shape
            in_two = (B, T * D, D)         # row (t*B + b) of the GEMM <-> inp[b, t]
        else:
            T, B, D = inp.shape
            in_two = (0, D, 0)
        dev = inp.device
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        Cs = torch.empty(T, B, 2, H, **f32)
        w_pk = None
        if pre is not None and pre.G is not None:
            G, pre.G = pre.G, None
        else:
            # padded frames of G stay unwritten: the recurrence kernel never uses them and
            # overwrites them with the zeros its backward expects
            G = torch.empty(T, B, 2, 4 * H, **f32)
            rows, nrows = (rl.rows, rl.n) if rl is not None else (None, T * B)
            dw, db1, db2 = ops.elem_delta(w_ih_f, w_ih_r), ops.elem_delta(b_ih_f, b_ih_r), ops.elem_delta(b_hh_f, b_hh_r)
            if MERGE_DIRECTIONS and None not in (dw, db1, db2):
                # both directions in one launch: the reverse direction's weights and biases are
                # reached through the two-level row / bias maps, nothing is concatenated.  At sizes where asr_sgemm has
                # a kernel for it, the weights go in PRE-SPLIT ({hi:lo} words of both directions, one small launch):
                # every one of the M / 128 row tiles would otherwise split the same weights again; the backward's dX
                # product reads the same copy
                if PACK_WEIGHTS and ops.packed_ok(nrows, 8 * H):
                    w_pk = ops.pack_hilo(w_ih_f, w_ih_r)
                ops.sgemm(inp, w_ih_f, G, nrows, 8 * H, D, a_two=in_two, b_two=(4 * H, D, dw), c_stride=8 * H,
                          bias1=b_ih_f, bias2=b_hh_f, bias_two=(4 * H, db1, db2), m_idx=rows,
                          b_pk=None if w_pk is None else (w_pk, D), a_packed=inp_packed)
            else:
                for d, (w_ih, b_ih, b_hh) in enumerate(((w_ih_f, b_ih_f, b_hh_f), (w_ih_r, b_ih_r, b_hh_r))):
                    ops.sgemm(inp, w_ih, G, nrows, 4 * H, D, a_two=in_two, b_stride=D, c_stride=8 * H,
                              c_off=d * 4 * H, bias1=b_ih, bias2=b_hh, m_idx=rows)
        # [2][4H][H]: nn.LSTM keeps the two directions' recurrent weights in separate tensors, the kernels read one
        W_hh = whh_cat if whh_cat is not None else torch.cat([w_hh_f, w_hh_r], 0)
        if last:
            Hout = torch.empty(B, T, 2 * H, **f32)
            ho_t, ho_b = 2 * H, T * 2 * H
        else:
            Hout = torch.empty(T, B, 2 * H, **f32)
            ho_t, ho_b = B * 2 * H, 2 * H
        ws = _persist_ws(lib, B, H, dev)
        wsp = 0 if ws is None else ws.data_ptr()
        nch = PIPE_CHUNKS if (streams.ENABLED and T >= PIPE_MIN_T) else 1
        out, mask = Hout, None
        if nxt is None or last or nch <= 1:
            check(lib.asr_blstm_fwd(W_hh.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), ho_t, ho_b,
                                    lengths_dev.data_ptr(), T, B, H, wsp, st))
            assert nxt is None, "a layer is only asked to prepare its successor when pipelining is on"
        else:
            p, seed = nxt.p, nxt.seed
            W_n = torch.cat(nxt.w, 0)                                            # [8H][2H]
            b_n = torch.cat([nxt.b[0] + nxt.b[1], nxt.b[2] + nxt.b[3]], 0)      # [8H]
            G_n = torch.empty(T, B, 2, 4 * H, **f32)
            if p > 0.0:
                out = torch.empty_like(Hout)
                mask = torch.empty(Hout.numel(), device=dev, dtype=torch.uint8)
            pipe = streams.pipe_stream(dev)
            main = torch.cuda.current_stream(dev)
            for t_ in (Hout, out, mask, G_n, W_n, b_n):
                if t_ is not None:
                    t_.record_stream(pipe)
            pipe.wait_stream(main)
            with torch.cuda.stream(pipe):
                G_n.view(T * B, 8 * H).copy_(b_n.expand(T * B, 8 * H))
            for s0, s1 in _chunks(T, nch):
                check(lib.asr_blstm_fwd_range(W_hh.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), ho_t,
                                              ho_b, lengths_dev.data_ptr(), T, B, H, s0, s1, wsp, st))
                pipe.wait_stream(main)
                with torch.cuda.stream(pipe):
                    for d in range(2):
                        t0, t1 = (s0, s1) if d == 0 else (T - s1, T - s0)
                        rows = (t1 - t0) * B
                        off = t0 * B * 2 * H + d * H
                        if p > 0.0:
                            check(lib.asr_dropout_fwd_block(Hout.data_ptr() + 4 * off, out.data_ptr() + 4 * off,
                                                            mask.data_ptr() + off, rows, H, 2 * H, off,
                                                            C.c_float(p), C.c_uint64(seed), _stream()))
                        # G_next[t0:t1] += X[t0:t1, :, d-th half] W_next[:, d-th half]^T
                        ops.sgemm(out, W_n, G_n, rows, 8 * H, H, a_stride=2 * H, a_off=off, b_stride=2 * H,
                                  b_off=d * H, c_stride=8 * H, c_off=t0 * B * 8 * H, accumulate=True)
            main.wait_stream(pipe)
            nxt.G = G_n
        ctx.save_for_backward(inp, G, Cs, Hout, lengths_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r)
        ctx.w_pk = w_pk
        ctx.inp_packed = inp_packed
        ctx.W_hh = W_hh            # a copy of the weights as they are now: the backward reads it instead of concatenating again
        ctx.mask = mask
        ctx.wparams = (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        ctx.p = nxt.p if mask is not None else 0.0
        ctx.dims = (B, T, D, H, first, in_two, ho_t, ho_b)
        ctx.rl = rl
        ctx.in_keep, ctx.in_scale = in_keep, in_scale
        ctx.in_p = in_drop[0] if in_keep is not None else 0.0
        return out

    @staticmethod
    def backward(ctx, dOut):
        lib = _lib.load()
        if getattr(ctx, "consumed", False):
            # the backward overwrites the saved gates with their gradients in place (raw pointers:
            # autograd's version counters cannot see it), so a second pass would run on destroyed data
            raise RuntimeError("asr_amd.Encoder: backward through a BLSTM layer a second time -- its saved "
                               "activations are consumed in place by the first pass (retain_graph is not supported)")
        ctx.consumed = True
        B, T, D, H, first, in_two, ho_t, ho_b = ctx.dims
        inp, G, Cs, Hout, lengths_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r = ctx.saved_tensors
        rl = ctx.rl
        dev = inp.device
        st = _stream()
        dOut = dOut.contiguous()                   # same layout as Hout
        f32 = dict(device=dev, dtype=torch.float32)
        if ctx.mask is not None:                   # the dropout this layer applied to its own output
            dHout = torch.empty_like(dOut)
            check(lib.asr_dropout_bwd(dOut.data_ptr(), ctx.mask.data_ptr(), dHout.data_ptr(), dOut.numel(),
                                      C.c_float(ctx.p), st))
        else:
            dHout = dOut
        dc_ws = torch.empty(2 * B * H, **f32)
        W_hh = ctx.W_hh.view(2, 4 * H, H)
        ws = _persist_ws(lib, B, H, dev)
        wsp = 0 if ws is None else ws.data_ptr()
        # [2][H][4H] for the step-wise kernels; the reduce-scatter form reads W_hh only: no transposition on its chain
        W_hhT = ops.transpose(W_hh) if lib.asr_blstm_bwd_needs_transposed(T, B, H, ho_t, ho_b, int(ws is not None)) else W_hh
        dG = G.view(T * B, 8 * H)                  # becomes d loss / d pre-activation gates (0 at padding)
        need_dx = ctx.needs_input_grad[0]
        nch = PIPE_CHUNKS if (streams.ENABLED and PIPE_BWD and T >= PIPE_MIN_T and need_dx and not first) else 1
        dIn = None
        if nch <= 1:
            check(lib.asr_blstm_bwd(W_hh.data_ptr(), W_hhT.data_ptr(), G.data_ptr(), Cs.data_ptr(), dHout.data_ptr(),
                                    ho_t, ho_b, lengths_dev.data_ptr(), dc_ws.data_ptr(), T, B, H, wsp, st))
            # critical path first: the input gradient feeds the next layer's recurrence
            if need_dx:
                # dX = dG[:, fwd] W_ih_f + dG[:, rev] W_ih_r, written in the input's own layout
                # with a row list the padded frames of dX are not written: the layer below never
                # reads them (its recurrence masks by length); the caller of the first layer might
                rows, nrows = (rl.rows, rl.n) if rl is not None else (None, T * B)
                dIn = torch.zeros_like(inp) if (first and rl is not None) else torch.empty_like(inp)
                keep, ks = ctx.in_keep, ctx.in_scale      # dropout of the input: its backward rides the epilogue
                dw = ops.elem_delta(w_ih_f, w_ih_r)
                if MERGE_DIRECTIONS and dw is not None:
                    ops.sgemm(dG, w_ih_f, dIn, nrows, D, 8 * H, a_stride=8 * H, b_mode=1, b_two=(4 * H, D, dw),
                              c_two=in_two, m_idx=rows, keep=keep, keep_scale=ks,
                              b_pk=None if ctx.w_pk is None else (ctx.w_pk, D))
                else:
                    for d, w_ih in enumerate((w_ih_f, w_ih_r)):
                        ops.sgemm(dG, w_ih, dIn, nrows, D, 4 * H, a_stride=8 * H, a_off=d * 4 * H, b_mode=1,
                                  b_stride=D, c_two=in_two, accumulate=(d == 1), m_idx=rows,
                                  keep=keep if d == 1 else None, keep_scale=ks)
        else:
            # time-major input (T,B,D): the rows of a frame range are contiguous in dG and dX
            dIn = torch.empty_like(inp)
            pipe = streams.pipe_stream(dev)
            main = torch.cuda.current_stream(dev)
            for t_ in (G, dIn):
                t_.record_stream(pipe)
            pipe.wait_stream(main)
            with torch.cuda.stream(pipe):
                dIn.zero_()
            for s0, s1 in _chunks(T, nch):
                check(lib.asr_blstm_bwd_range(W_hh.data_ptr(), W_hhT.data_ptr(), G.data_ptr(), Cs.data_ptr(),
                                              dHout.data_ptr(), ho_t, ho_b, lengths_dev.data_ptr(),
                                              dc_ws.data_ptr(), T, B, H, s0, s1, wsp, st))
                pipe.wait_stream(main)
                with torch.cuda.stream(pipe):
                    for d, w_ih in enumerate((w_ih_f, w_ih_r)):
                        t0, t1 = (T - s1, T - s0) if d == 0 else (s0, s1)
                        ops.sgemm(dG, w_ih, dIn, (t1 - t0) * B, D, 4 * H, a_stride=8 * H,
                                  a_off=t0 * B * 8 * H + d * 4 * H, b_mode=1, b_stride=D, c_stride=D,
                                  c_off=t0 * B * D, accumulate=True)
            main.wait_stream(pipe)
            if ctx.in_keep is not None:
                check(lib.asr_dropout_bwd(dIn.data_ptr(), ctx.in_keep.data_ptr(), dIn.data_ptr(), dIn.numel(),
                                          C.c_float(ctx.in_p), st))
        # weight / bias gradients: nothing downstream reads them -> side stream, overlapping the
        # next layer's (latency-bound) recurrence kernel
        hout_two = (B, ho_b, ho_t)                 # row (t*B + b) <-> Hout element offset
        side_reads = (G, inp, Hout) + ((rl.rows,) if rl is not None else ())    # rows and prev share a block

        def grad_w_hh(d):
            # dW_hh[d] = sum_t dG_t^T h_{t-1}: forward pairs (t, t-1), reverse pairs (t, t+1);
            # padded rows of dG and Hout are zero, so ragged lengths need no masking here
            # (the row lists only skip their work)
            gW = torch.empty(4 * H, H, **f32)
            if T > 1:
                K = (T - 1) * B
                kl = None
                if rl is not None:
                    kl, K = (rl.prev, rl.n_prev) if d == 0 else (rl.rows, rl.n_next)
                sk = ops.pick_splitk(4 * H, H, K)
                if sk > 1:
                    gW.zero_()
                a_off = d * 4 * H + (B * 8 * H if d == 0 else 0)
                b_off = d * H + (0 if d == 0 else ho_t)
                ops.sgemm(dG, Hout, gW, 4 * H, H, K, a_mode=1, a_stride=8 * H, a_off=a_off, b_mode=1,
                          b_two=hout_two, b_off=b_off, c_stride=H, splitk=sk, k_idx=kl)
            else:
                gW.zero_()
            return gW

        # The bottom layer's gradients are the one piece of side-stream work nothing hides: no recurrence follows it and
        # the main stream would sit idle until the join.  With no input gradient to compute there, the main stream takes
        # the two dW_hh products while the side stream does dW_ih and the bias sums (two queues fill the chip better
        # than one: these are 30-50 tile split-K launches).
        split_streams = first and not need_dx and streams.ENABLED and not any(p.grad is not None for p in ctx.wparams)
        with streams.on_side(dev, *side_reads, params=ctx.wparams):
            db = torch.empty(8 * H, **f32)
            ops.colsum(dG, db)                     # b_ih and b_hh receive the same column sums
            if ws is not None:                     # a timed-out wait in this backward: NaN gradients NOW, not garbage
                persist.poison_from(ws, db)
            # dW_ih of BOTH directions in one product: dG^T (8H x K) @ X (K x D) -> [2][4H][D]
            gWi = torch.empty(2, 4 * H, D, **f32)
            kl, K = (rl.rows, rl.n) if rl is not None else (None, T * B)
            sk = ops.pick_splitk(8 * H, D, K)
            if sk > 1:
                gWi.zero_()
            ops.sgemm(dG, inp, gWi, 8 * H, D, K, a_mode=1, a_stride=8 * H, b_mode=1, b_two=in_two, c_stride=D,
                      splitk=sk, k_idx=kl, b_packed=ctx.inp_packed)
            gW_hh = None if split_streams else [grad_w_hh(d) for d in range(2)]
            db2 = db.clone()       # b_hh's copy (the clone reads db: it belongs on the stream that wrote it)
        if split_streams:          # enqueued AFTER the side block: its entry event must not wait for these two
            gW_hh = [grad_w_hh(d) for d in range(2)]
        grads = {d: (gWi[d], gW_hh[d], db[d * 4 * H:(d + 1) * 4 * H], db2[d * 4 * H:(d + 1) * 4 * H]) for d in range(2)}
        ctx.W_hh = ctx.w_pk = None
        return (dIn, None, None, None, None, None, None, None, None, None) + grads[0] + grads[1]


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
        if getattr(hp, "encoder_type", None) == "CNN" and hp.frame_stacking <= 1:
            # Models/encoder.py:18-22: without frame stacking this selects the VGG front-end (CNN_embedding :38-62),
            # which SURVEY.md section 2 row 13 leaves outside the path -- refuse instead of silently building a plain BLSTM
            raise NotImplementedError("asr_amd.Encoder: encoder_type='CNN' (the VGG front-end of Models/encoder.py:38-62) "
                                      "is outside the accelerated path")
        self.num_layers = hp.num_encoder_layer
        self.hidden = hp.num_hidden_nodes_encoder
        self.dropout = float(hp.encoder_dropout)
        self.bi_lstm = BiLSTMParams(hp.lmfb_dim * hp.frame_stacking, self.hidden, self.num_layers)
        self._step = 0

    def forward(self, x, lengths, lengths_host=None):
        """lengths_host: the same lengths on the CPU when `lengths` was moved to the device already
        (AttModel does that once for both halves); lets the GEMMs skip the padded frames."""
        if not x.is_cuda:
            raise RuntimeError("asr_amd.Encoder: input must be on the GPU; the HIP path has no CPU fallback")
        lengths_dev = lengths_to_device(lengths, x.device)
        rl = _row_lists(lengths if lengths_host is None else lengths_host, x.shape[1], x.shape[0], x.device)
        p = self.dropout if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0
        h = x.float()
        flat = self.bi_lstm.flat()
        T = h.shape[1]
        pipelined = streams.ENABLED and PIPE_FWD and PIPE_CHUNKS > 1 and T >= PIPE_MIN_T
        pre = None
        in_drop = None        # dropout between layers: applied by the consuming layer (fused backward)
        # (tried: concatenating all layers' [2][4H][H] recurrent weights on the side stream ahead of the layers -- the
        # event wait costs what the 6 us launches save: 17.07 vs 17.02 ms per step; each layer concatenates its own)
        whh = [None] * self.num_layers
        for layer in range(self.num_layers):
            last = layer == self.num_layers - 1
            nxt = None
            if pipelined and not last:
                n = flat[8 * (layer + 1): 8 * (layer + 1) + 8]     # (w_ih, w_hh, b_ih, b_hh) x (fwd, rev)
                nxt = _Next(n[0], n[2], n[3], n[4], n[6], n[7], p, seed + 7919 * layer)
            h = _BLSTMLayerFunction.apply(h, lengths_dev, rl, layer == 0, last, self.hidden, pre, nxt, in_drop, whh[layer],
                                          *flat[8 * layer: 8 * layer + 8])
            in_drop = (p, seed + 7919 * layer) if (nxt is None and p > 0.0 and not last) else None
            pre = nxt
        return h
