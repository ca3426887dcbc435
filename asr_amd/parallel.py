"""Data-parallel gradient exchange: bucketed all-reduce on RCCL, overlapped with backward.

The reference's only collective is the gradient all-reduce hidden inside
``DistributedDataParallel`` (train.py:141, NCCL, mean over ranks).  Stock DDP
still works on the ``asr_amd`` module (the drop-in path); this reducer is the
MI355X-first version: buckets follow the order in which the HIP backward
finishes gradients -- decoder+attention first (final when the decoder BPTT
ends, overlapping the whole encoder backward), then encoder layers top-down --
and each bucket is one flat fp32 buffer, so every all-reduce is a single large
RCCL call (xGMI is point-to-point: few large messages beat many small ones).
The 1/world_size of DDP's mean is applied with asr_scale when the bucket lands.

Set ``TORCH_NCCL_HIGH_PRIORITY=1`` before ``init_process_group``: HIP multiplexes a process's
streams onto a few hardware queues, and at normal priority RCCL's internal stream lands on the
queue of the main stream, so a bucket's wait for its weight gradients (side stream) and the
collective itself hold back the next layer's recurrence kernel queued behind them (measured in
a group of one on MI355X: 37.8 ms per step without, 34.2 with; 33.4 with no reducer at all).

Differences from DistributedDataParallel, on purpose and checked:
* one pass per ``finish()``: a second ``backward()`` before ``finish()`` raises (DDP would reduce each
  pass); accumulate micro-batches under ``no_sync()`` and let the last backward run outside it;
* a parameter that received no gradient in the pass is reduced as zeros (every rank issues the same
  collectives in the same order; DDP raises unless ``find_unused_parameters`` is set);
* buffers (the Conformer's BatchNorm running statistics) are broadcast once by
  ``broadcast_parameters()``; DDP re-broadcasts them in every forward.  They do not enter the
  training-mode arithmetic; call ``broadcast_buffers()`` before a checkpoint or an evaluation pass
  if rank 0's statistics are to win, as they do under DDP;
* ``close()`` before ``dist.destroy_process_group()``: it removes the hooks, waits for and drops the
  side streams that shared hardware queues with RCCL's and restores ``streams.ENABLED``.
"""
import contextlib
import os
import warnings

import torch
import torch.distributed as dist

from . import ops, streams


CONFORMER_LAYERS_PER_BUCKET = 4


class BucketedGradReducer:
    def __init__(self, model, process_group=None, force=False):
        """force: run the bucket machinery (hooks, packing, collective, mean) even in a group of
        one -- a single-GPU rehearsal of the exact stream choreography the N-GPU job uses."""
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        if (self.active and dist.get_backend(process_group) == "nccl"
                and os.environ.get("TORCH_NCCL_HIGH_PRIORITY", "0") not in ("1", "true", "True")):
            warnings.warn("asr_amd.BucketedGradReducer: TORCH_NCCL_HIGH_PRIORITY=1 was not set before "
                          "init_process_group; RCCL will share the main stream's hardware queue and the "
                          "overlap with the backward is lost (see asr_amd/parallel.py)")
        # this reducer is stream-aware (see _launch): the weight gradients may be produced on the side stream
        self._streams_was = streams.ENABLED
        streams.enable(True)
        enc, dec = model.encoder, model.decoder
        self.buckets = []                       # list of lists of parameters, in backward-completion order
        self.buckets.append(list(dec.parameters()))
        if hasattr(enc, "bi_lstm"):
            n_layers = enc.num_layers
            for layer in range(n_layers - 1, -1, -1):
                names = [f"{k}_l{layer}{sfx}" for sfx in ("", "_reverse")
                         for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                self.buckets.append([getattr(enc.bi_lstm, n) for n in names])
        else:
            # Conformer encoder (Models/Transformer_encoder.py): layers top-down, CONFORMER_LAYERS_PER_BUCKET
            # to a bucket (11.6 MB of gradients per layer at d_model 256 -> 46 MB messages), then the subsampling
            layers = list(enc.encoder.layers)
            for hi in range(len(layers), 0, -CONFORMER_LAYERS_PER_BUCKET):
                group = layers[max(0, hi - CONFORMER_LAYERS_PER_BUCKET):hi]
                self.buckets.append([p for layer in reversed(group) for p in layer.parameters()])
            self.buckets.append(list(enc.cnn_encoder.parameters()))
        self.flat = [None] * len(self.buckets)
        self.work = [None] * len(self.buckets)
        self._hooks = []
        self._counts = [0] * len(self.buckets)
        self._where = {}
        self._sync = True
        self._closed = False
        for bi, ps in enumerate(self.buckets):
            for p in ps:
                self._where[p] = bi
                if self.active:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ set-up / tear-down --
    def broadcast_parameters(self, src=0):
        """One-time parameter (and buffer) broadcast from rank 0 (what DDP does at wrap time)."""
        if not self.active:
            return
        for p in self.model.parameters():
            dist.broadcast(p.data, src=src, group=self.group)
        self.broadcast_buffers(src)

    def broadcast_buffers(self, src=0):
        """BatchNorm running statistics of the Conformer encoder (DDP does this in every forward)."""
        if not self.active:
            return
        for b in self.model.buffers():
            dist.broadcast(b.data, src=src, group=self.group)

    def close(self):
        """Undo everything the constructor did to the process: remove the gradient hooks, wait for
        outstanding collectives, wait for and DROP the cached side / pipe streams (they were created at
        high priority to get a hardware queue of their own next to RCCL's streams; a later process
        group -- evaluation, resume -- gets fresh ones) and restore ``streams.ENABLED``.  Call it
        before ``dist.destroy_process_group()``.  Idempotent."""
        if self._closed:
            return
        self._closed = True
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for bi, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[bi] = None
        dev = next((p.device for ps in self.buckets for p in ps if p.is_cuda), None)
        if dev is not None:
            torch.cuda.current_stream(dev).synchronize()
            streams.reset(dev)
        self.flat = [None] * len(self.buckets)
        self.active = False
        streams.enable(self._streams_was)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (DDP.no_sync): backward passes inside the block only accumulate into
        ``.grad``; the first backward outside it reduces the sums."""
        was, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = was

    # ----------------------------------------------------------------------------- per step --
    def _views(self, bi):
        """The bucket's flat buffer as one view per parameter (same shapes as the gradients)."""
        flat, out, o = self.flat[bi], [], 0
        for p in self.buckets[bi]:
            out.append(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        return out

    def _on_grad(self, p):
        if not self._sync:
            return
        bi = self._where[p]
        if self._counts[bi] >= len(self.buckets[bi]):
            raise RuntimeError("asr_amd.BucketedGradReducer: a second backward() reached a bucket that was already "
                               "reduced in this step; call finish() after every backward(), or accumulate "
                               "micro-batches under no_sync()")
        self._counts[bi] += 1
        if self._counts[bi] == len(self.buckets[bi]):
            self._launch(bi)

    def _launch(self, bi):
        ps = self.buckets[bi]
        n = sum(p.numel() for p in ps)
        if self.flat[bi] is None or self.flat[bi].numel() != n:
            self.flat[bi] = torch.empty(n, device=ps[0].device, dtype=torch.float32)
        flat = self.flat[bi]

        def pack_and_reduce():
            # one multi-tensor copy per bucket, not one launch per parameter
            torch._foreach_copy_(self._views(bi), [p.grad for p in ps])
            # async_op: RCCL runs on the process group's own stream, ordered after the kernels
            # enqueued so far on the CURRENT stream; the backward keeps issuing work meanwhile
            self.work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        if ps[0].is_cuda and streams.ENABLED:
            # The weight gradients are produced on the side stream (asr_amd/streams.py): pack and
            # launch the bucket THERE, so the main stream -- the latency-bound recurrence chain --
            # never waits for them.  The side stream first catches up with the main stream (the few
            # gradients the main stream produced, and `flat`'s previous use in finish()).
            dev = ps[0].device
            side = streams.side_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                pack_and_reduce()
        else:
            pack_and_reduce()

    def finish(self):
        """Wait for every bucket, apply DDP's mean and scatter back into ``.grad``.  Every rank issues
        every bucket's collective, in bucket order: a parameter without a gradient is sent as zeros."""
        if not self.active:
            return
        for bi, ps in enumerate(self.buckets):
            if self.work[bi] is None:
                for p in ps:       # hooks did not fire for (some of) the bucket: unused parameter, grads set by hand
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(bi)
            self.work[bi].wait()
            flat = self.flat[bi]
            if flat.is_cuda:
                ops.scale_(flat, 1.0 / self.world)
            else:
                flat.mul_(1.0 / self.world)     # gloo/CPU rehearsal of the same bucket logic
            torch._foreach_copy_([p.grad for p in ps], self._views(bi))
            self.work[bi] = None
            self._counts[bi] = 0

    def bytes_per_step(self):
        return 4 * sum(p.numel() for ps in self.buckets for p in ps)

    def allreduce_probe(self, iters=10):
        """Times each bucket's all-reduce alone (HIP events round the collective, `iters` back-to-back
        calls) and returns per-bucket bytes, seconds, algorithm GB/s and bus GB/s (algorithm bandwidth
        x 2(N-1)/N, the per-link load of a ring; compare with 153 GB/s per xGMI link).  Collective on
        every rank: call it on all of them."""
        out = []
        if not self.active:
            return out
        for bi, ps in enumerate(self.buckets):
            n = sum(p.numel() for p in ps)
            buf = torch.zeros(n, device=ps[0].device, dtype=torch.float32)
            dist.all_reduce(buf, group=self.group)
            if buf.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(iters):
                    dist.all_reduce(buf, group=self.group)
                e1.record()
                torch.cuda.synchronize()
                sec = e0.elapsed_time(e1) * 1e-3 / iters
            else:
                import time
                t0 = time.perf_counter()
                for _ in range(iters):
                    dist.all_reduce(buf, group=self.group)
                sec = (time.perf_counter() - t0) / iters
            alg = 4.0 * n / sec / 1e9
            out.append({"bucket": bi, "bytes": 4 * n, "us": sec * 1e6, "alg_GBs": alg,
                        "bus_GBs": alg * 2.0 * (self.world - 1) / max(1, self.world)})
        return out
