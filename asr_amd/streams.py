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


# How the helper streams are made (ASR_SIDE_STREAM_KIND):
#   "normal" (DEFAULT): torch.cuda.Stream() -- one of the 4 normal-priority hardware queues.  Created by
#       prepare() BEFORE the process group exists it owns its queue and overlaps the backward; created
#       after RCCL's streams it shares the MAIN stream's queue and the overlap disappears (37.8 vs 35.4 ms
#       per step; no error).  BucketedGradReducer warns when it finds the second situation.
#   "cumask": a stream created with a compute-unit mask (asr_stream_create) -- HIP gives such
#       a stream a hardware queue of its OWN, at normal priority.  ASR_SIDE_CU_MASK picks the CUs: "all"
#       (default) or "hi:N" = the upper N compute units of every XCD (CU-mask bit i is compute unit i // 8
#       of XCD i % 8 on MI355X), which keeps the weight-gradient GEMMs off N..31 only.  Measured slower
#       than "normal" with every mask: A/B runs only.
#   "priority": torch.cuda.Stream(priority=-1), round 1's way to a queue of its own.  A high-priority
#       stream that waits on an event of the normal-priority main stream can starve it: with the stream
#       created after RCCL's, the main stream's recurrence kernel never ran (tools/stall_probe.py,
#       "plain" order) -- kept for A/B runs only.
KIND = os.environ.get("ASR_SIDE_STREAM_KIND", "normal")
_streams = {}
_pipes = {}
_owned = []        # raw handles of the cumask streams: they live as long as the process


def _cu_mask_words():
    spec = os.environ.get("ASR_SIDE_CU_MASK", "all")
    if spec == "all":
        return None
    kind, n = spec.split(":")
    n = int(n)
    bits = 0
    for cu in range(32):
        on = cu >= 32 - n if kind == "hi" else cu < n
        if on:
            for xcd in range(8):
                bits |= 1 << (cu * 8 + xcd)
    return [(bits >> (32 * w)) & 0xffffffff for w in range(8)]


def _make(device):
    if KIND == "priority":
        return torch.cuda.Stream(device=device, priority=-1)
    if KIND == "normal":
        return torch.cuda.Stream(device=device)
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    words = _cu_mask_words()
    handle = C.c_void_p()
    for _ in range(int(os.environ.get("ASR_SIDE_SPACERS", "0")) if not _owned else 0):   # experiment: queue placement
        spacer = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.asr_stream_create(C.byref(spacer), None, 0))
    with torch.cuda.device(device):
        if words is None:
            _lib.check(lib.asr_stream_create(C.byref(handle), None, 0))
        else:
            arr = (C.c_uint32 * len(words))(*words)
            _lib.check(lib.asr_stream_create(C.byref(handle), C.cast(arr, C.c_void_p), len(words)))
    _owned.append((device.index, handle.value))
    return torch.cuda.ExternalStream(handle.value, device=device)


def prepared(device):
    """True once this device's helper streams exist (prepare(), or first use)."""
    return device.index in _streams


def prepare(device):
    """Create the helper streams NOW and push one trivial operation through each, so that they own
    their hardware queues before anybody else's streams exist.  Call it before
    ``dist.init_process_group``: HIP hands the first four streams of a priority class a hardware queue
    each and lets later ones share the least-used queue -- after RCCL has created its streams that is
    the main stream's, and work meant to overlap the backward then queues behind it (no error, no
    deadlock: only the overlap is lost)."""
    for s in (side_stream(device), pipe_stream(device)):
        with torch.cuda.stream(s):
            torch.zeros(1, device=device)
        s.synchronize()


def side_stream(device):
    s = _streams.get(device.index)
    if s is None:
        s = _make(device)
        _streams[device.index] = s
    return s


def pipe_stream(device):
    """Second helper stream: the chunk-pipelined GEMMs that ARE on the critical path (the main
    stream joins it at the end of every layer), kept apart from the weight-gradient stream so the
    two kinds of work never queue behind each other."""
    s = _pipes.get(device.index)
    if s is None:
        s = _make(device)
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
    """Wait for the helper streams' work.  Pool streams of the "priority" / "normal" kinds are then
    dropped, so the next user draws fresh ones; a CU-masked stream owns its hardware queue, shares it
    with nobody and is kept (destroying it would pull the handle from under the caching allocator,
    which records an event on every stream a block was used on when the block is freed).
    BucketedGradReducer.close() calls this before its process group is destroyed."""
    for cache in (_streams, _pipes):
        for idx in list(cache):
            if device is None or device.index == idx:
                cache[idx].synchronize()
                if KIND != "cumask":
                    cache.pop(idx)


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
