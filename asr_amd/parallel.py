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

Stream priorities: everything runs at NORMAL priority, and ``TORCH_NCCL_HIGH_PRIORITY`` must stay
unset.  HIP multiplexes a process's streams onto four hardware queues per priority class; a
high-priority stream whose next packet is a wait on an event of a normal-priority stream can keep
that stream's queue from being served at all -- measured on MI355X / ROCm 7.2 as a step that never
ends, twice: a priority -1 side stream created after RCCL's streams waiting on the main stream's
recurrence (tools/stall_probe.py, "plain" order), and RCCL's high-priority stream waiting on a
normal-priority side stream at the config-2 size (bench.py, ASR_BENCH_FORCE_DP=1).  With all streams
normal the only thing to lose is the overlap, when the side stream has to share the main stream's
queue: call ``asr_amd.streams.prepare(device)`` before ``dist.init_process_group`` so the helper
streams own their queues before RCCL creates its streams (group of one on MI355X: 34.2 ms per step
with the reducer, 33.6 without any).

Differences from DistributedDataParallel, on purpose and checked:
* one pass per ``finish()``: a second ``backward()`` before ``finish()`` raises (DDP would reduce each
  pass); accumulate micro-batches under ``no_sync()`` and let the last backward run outside it;
* a parameter that received no gradient in the pass is reduced as zeros (DDP raises unless
  ``find_unused_parameters`` is set).  Collectives are issued in BUCKET ORDER on every rank, whatever
  order the gradients arrive in: a bucket whose gradients are complete is launched from the backward
  only once every bucket before it has been launched, otherwise it waits for ``finish()``, which
  launches the rest in order.  Ranks whose sets of unused parameters differ therefore still issue the
  same sequence of all-reduces (tests/test_dp_gloo.py::test_unused_parameter_on_one_rank_only).  The
  zero fill leaves ``p.grad`` a tensor of zeros (after the mean: the other ranks' average), not None:
  an optimizer that skips ``grad is None`` parameters will now update them -- Adam applies its weight
  decay and decays the moments of a parameter it would have left alone;
* buffers (the Conformer's BatchNorm running statistics) are broadcast once by
  ``broadcast_parameters()``; DDP re-broadcasts them in every forward.  They do not enter the
  training-mode arithmetic; call ``broadcast_buffers()`` before a checkpoint or an evaluation pass
  if rank 0's statistics are to win, as they do under DDP;
* ``close()`` before ``dist.destroy_process_group()``: it removes the hooks (they would fire into a
  dead group), waits for the side streams and restores ``streams.ENABLED``.
"""
import contextlib
import os
import warnings

import torch
import torch.distributed as dist

from . import ops, streams


CONFORMER_LAYERS_PER_BUCKET = 4


class _HostStagedWork:
    """gloo has no device transport: handed a device tensor it stages through pinned memory on pool
    streams and worker threads of its own.  For the one-GPU rehearsal of the N-rank path (all ranks on
    one device over gloo) the reducer does that staging itself, where it can be seen: the packed
    bucket is copied to pinned host memory on the stream that packed it (asynchronous, event behind
    it); ``wait()`` -- called from finish(), in bucket order on every rank -- waits for the event,
    all-reduces the HOST tensor and copies the sums back on the current stream."""

    def __init__(self, flat, host, group):
        self.flat, self.host, self.group = flat, host, group
        self.host.copy_(flat, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(flat.device))

    def wait(self):
        self.ev.synchronize()
        dist.all_reduce(self.host, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.copy_(self.host, non_blocking=True)


class BucketedGradReducer:
    def __init__(self, model, process_group=None, force=False):
        """force: run the bucket machinery (hooks, packing, collective, mean) even in a group of
        one -- a single-GPU rehearsal of the exact stream choreography the N-GPU job uses."""
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        if (self.active and dist.get_backend(process_group) == "nccl"
                and os.environ.get("TORCH_NCCL_HIGH_PRIORITY", "0") in ("1", "true", "True")):
            warnings.warn("asr_amd.BucketedGradReducer: TORCH_NCCL_HIGH_PRIORITY=1 is set; a high-priority RCCL "
                          "stream that waits on this reducer's normal-priority streams can starve them and "
                          "the step never ends (see asr_amd/parallel.py) -- unset it")
        dev0 = next((p.device for p in model.parameters() if p.is_cuda), None)
        if (self.active and dev0 is not None and dist.get_backend(process_group) == "nccl"
                and streams.KIND == "normal" and not streams.prepared(dev0)):
            warnings.warn("asr_amd.BucketedGradReducer: the process group exists but asr_amd.streams.prepare(device) "
                          "has not run; helper streams created from now on share the main stream's hardware queue "
                          "and the weight-gradient / all-reduce overlap is lost (results are unaffected) -- call "
                          "streams.prepare(device) before dist.init_process_group")
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
        self.host = [None] * len(self.buckets)
        self.work = [None] * len(self.buckets)
        # gloo + device tensors: stage through pinned memory here (see _HostStagedWork);
        # ASR_GLOO_DEVICE_TENSORS=1 hands gloo the device tensors instead (its own staging)
        self._stage = (self.active and dist.get_backend(process_group) == "gloo"
                       and os.environ.get("ASR_GLOO_DEVICE_TENSORS", "0") != "1")
        self._hooks = []
        self._counts = [0] * len(self.buckets)
        self._next = 0                          # buckets 0 .. _next-1 have been launched in this step
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
            self._broadcast(p.data, src)
        self.broadcast_buffers(src)

    def broadcast_buffers(self, src=0):
        """BatchNorm running statistics of the Conformer encoder (DDP does this in every forward)."""
        if not self.active:
            return
        for b in self.model.buffers():
            self._broadcast(b.data, src)

    def _broadcast(self, t, src):
        if self._stage and t.is_cuda:
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=self.group)

    def close(self):
        """Undo everything the constructor did to the process: remove the gradient hooks, wait for
        outstanding collectives and for the side / pipe streams' work (streams.reset) and restore
        ``streams.ENABLED``.  Call it before ``dist.destroy_process_group()``.  Idempotent."""
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
        self.host = [None] * len(self.buckets)
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
        # one global launch order: bucket _next goes out as soon as it is complete, a later bucket
        # that completes first waits for its turn (or for finish())
        while self._next < len(self.buckets) and self._counts[self._next] == len(self.buckets[self._next]):
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        ps = self.buckets[bi]
        n = sum(p.numel() for p in ps)
        if self.flat[bi] is None or self.flat[bi].numel() != n:
            self.flat[bi] = torch.empty(n, device=ps[0].device, dtype=torch.float32)
        flat = self.flat[bi]

        def pack_and_reduce():
            # one multi-tensor copy per bucket, not one launch per parameter
            torch._foreach_copy_(self._views(bi), [p.grad for p in ps])
            if self._stage and flat.is_cuda:
                if self.host[bi] is None or self.host[bi].numel() != n:
                    self.host[bi] = torch.empty(n, dtype=torch.float32, pin_memory=True)
                self.work[bi] = _HostStagedWork(flat, self.host[bi], self.group)
                return
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
        for bi in range(self._next, len(self.buckets)):
            # not launched from the backward: hooks did not fire for (some of) the bucket -- unused parameter,
            # gradients set by hand -- or an earlier bucket was incomplete
            for p in self.buckets[bi]:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            self._launch(bi)
        self._next = 0
        for bi, ps in enumerate(self.buckets):
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
            if self._stage and buf.is_cuda:     # rehearsal over gloo: the wire time of the host tensors
                buf = torch.zeros(n, dtype=torch.float32, pin_memory=True)
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
