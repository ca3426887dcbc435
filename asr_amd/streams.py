"""Side stream for work that is off the backward's critical path.

The backward of the path is a chain of latency-bound kernels (the BLSTM and decoder
recurrences occupy a fraction of the CUs and mostly wait on each other); the weight
gradients, on the other hand, are large independent GEMMs that nothing in the rest of
the backward reads.  They are therefore enqueued on a second HIP stream, ordered after
the kernel that produces their input by an event, and run concurrently with the next
recurrence.  One callback at the end of the autograd pass makes the main stream wait
for the side stream, so every ``.grad`` is complete when ``backward()`` returns control
to the caller's stream.
"""
import os

import torch

# Off until a stream-aware owner of the gradients turns it on (BucketedGradReducer does; so can a
# training loop that only reads .grad after backward() has returned): anything that reads a
# gradient the moment autograd hands it over -- stock DistributedDataParallel's bucket hooks --
# would read it before the side stream has written it.  ASR_SIDE_STREAM=0 keeps it off (A/B timing).
ENABLED = False
_FORCED_OFF = os.environ.get("ASR_SIDE_STREAM", "1") == "0"


def enable(flag=True):
    global ENABLED
    ENABLED = bool(flag) and not _FORCED_OFF


_streams = {}


def side_stream(device):
    s = _streams.get(device.index)
    if s is None:
        # High priority on purpose: HIP multiplexes a process's streams onto few hardware queues, and
        # once RCCL has created its (high-priority) streams a normal-priority stream created here
        # lands on the SAME hardware queue as the main stream -- measured: the overlap vanishes
        # (37.8 vs 35.4 ms per step).  A high-priority stream keeps a queue of its own.
        s = torch.cuda.Stream(device=device, priority=-1)
        _streams[device.index] = s
    return s


_pipes = {}


def pipe_stream(device):
    """Second helper stream: the chunk-pipelined GEMMs that ARE on the critical path (the main
    stream joins it at the end of every layer), kept apart from the weight-gradient stream so the
    two kinds of work never queue behind each other."""
    s = _pipes.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device, priority=-1)
        _pipes[device.index] = s
    return s


def arm_join(device, main):
    """Queue the end-of-backward callback that makes `main` -- the stream this backward node runs
    on, captured now: the callback itself may run on another thread with another current stream --
    wait for the side stream.  Queued by every block (a few redundant event waits per pass) rather
    than once per pass: no cross-thread bookkeeping about which pass is running."""
    side = side_stream(device)
    torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_stream(side))


def wait_side(device):
    """Make the current stream wait for everything enqueued on the side stream so far."""
    if device.index in _streams:
        torch.cuda.current_stream(device).wait_stream(_streams[device.index])


def reset(device=None):
    """Drop the cached side / pipe streams (after waiting for their work): the next user gets fresh ones.
    BucketedGradReducer.close() calls this before the process group that shared their hardware queues
    is destroyed (asr_amd/parallel.py)."""
    for cache in (_streams, _pipes):
        for idx in list(cache):
            if device is None or device.index == idx:
                cache.pop(idx).synchronize()


class on_side:
    """with on_side(device, *tensors, params=...): ... -- run the block on the side stream, ordered
    after everything enqueued on the current stream so far; `tensors` were allocated on the main
    stream and are read by the block (their memory must not be recycled under it); `params` are
    the parameters whose gradients the block produces."""

    def __init__(self, device, *tensors, params=()):
        self.device = device
        self.tensors = tensors
        self.params = params
        self.ctx = None

    def __enter__(self):
        # a parameter that already holds a gradient gets the new one ADDED by autograd on the main
        # stream as soon as this backward node returns (gradient accumulation): stay on the main stream
        if not ENABLED or any(p.grad is not None for p in self.params):
            return self
        side = side_stream(self.device)
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        for t in self.tensors:
            if t is not None:
                t.record_stream(side)
        arm_join(self.device, main)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False
