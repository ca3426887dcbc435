"""CPU, world_size 2 over gloo: the bucketed gradient reducer averages gradients
like DDP (train.py:141) and keeps the backward-completion bucket order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, conformer=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=8, num_encoder_layer=2, lmfb_dim=4, num_classes=7, frame_stacking=1)
    if conformer:
        hp = R.make_hp(num_hidden_nodes_encoder=256, num_hidden_nodes_decoder=8, lmfb_dim=20, num_classes=7,
                       frame_stacking=1, encoder_type="Conformer")
    torch.manual_seed(100 + rank)                  # different weights per rank on purpose
    m = AttModel(hp)
    if conformer:
        import asr_amd.parallel as par
        m.encoder.encoder.layers = torch.nn.Sequential(*list(m.encoder.encoder.layers)[:6])   # 6 layers: buckets of 4 + 2
        assert par.CONFORMER_LAYERS_PER_BUCKET == 4
    red = BucketedGradReducer(m)
    red.broadcast_parameters()
    w0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    g = torch.Generator().manual_seed(rank)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    mine = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    red.finish()
    avg = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered) / world
    ws = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(ws, w0)
    ok = bool(torch.allclose(avg, want, atol=1e-6)) and all(torch.equal(ws[0], w) for w in ws)
    # bucket order: decoder first, then encoder layers top-down
    names = {p: n for n, p in m.named_parameters()}
    order = [names[ps[0]] for ps in red.buckets]
    if conformer:
        ok = ok and len(order) == 4 and order[0].startswith("decoder.") and order[1].startswith("encoder.encoder.layers.5.") \
            and order[2].startswith("encoder.encoder.layers.1.") and order[3].startswith("encoder.cnn_encoder.")
        bs = [torch.cat([b.detach().reshape(-1).float() for b in m.buffers()])]
        allb = [torch.zeros_like(bs[0]) for _ in range(world)]
        dist.all_gather(allb, bs[0])
        ok = ok and all(torch.equal(allb[0], b) for b in allb)
    else:
        ok = ok and order[0].startswith("decoder.") and order[1].endswith("_l1") and order[2].endswith("_l0")
    ok = ok and red.bytes_per_step() == 4 * sum(p.numel() for p in m.parameters())
    out[rank] = ok
    dist.destroy_process_group()


def test_reducer_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_reducer_world2_gloo_conformer():
    """The same reducer on AttModel(encoder_type='Conformer'): decoder, then Conformer layers top-down four to a
    bucket, then the subsampling; buffers (BatchNorm statistics) broadcast with the parameters."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, True), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def _semantics_worker(rank, world, port, out):
    """Hooks driven by a real autograd pass (a surrogate loss over the parameters: the HIP forward
    needs a GPU), so the bucket counting, the re-entry guard, no_sync, zero-filled unused
    parameters and close() are exercised as a training job would."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asr_amd import streams
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=8, num_encoder_layer=2, lmfb_dim=4, num_classes=7, frame_stacking=1)
    torch.manual_seed(7)
    m = AttModel(hp)
    params = list(m.parameters())
    was = streams.ENABLED
    red = BucketedGradReducer(m)
    red.broadcast_parameters()
    ok = red.active and streams.ENABLED

    def backward(ps=params):
        ((rank + 1.0) * sum((p * p).sum() for p in ps)).backward()

    def zero():
        for p in params:
            p.grad = None

    mean_scale = sum(2.0 * (r + 1) for r in range(world)) / world     # d/dp of (r+1) p^2, averaged over ranks
    # (1) one pass: every bucket launched from the hooks, finish() leaves the mean
    zero()
    backward()
    ok = ok and all(w is not None for w in red.work)
    red.finish()
    ok = ok and all(torch.allclose(p.grad, mean_scale * p.detach(), atol=1e-6) for p in params)
    # (2) a second backward before finish() must raise, not reduce a stale bucket
    zero()
    backward()
    try:
        backward()
        ok = False
    except RuntimeError as e:
        ok = ok and "no_sync" in str(e)
    red.finish()
    # (3) accumulation under no_sync(): two micro-batches, one reduction of their sum
    zero()
    with red.no_sync():
        backward()
        ok = ok and all(w is None for w in red.work)
    backward()
    red.finish()
    ok = ok and all(torch.allclose(p.grad, 2 * mean_scale * p.detach(), atol=1e-5) for p in params)
    # (4) a parameter that got no gradient is reduced as zeros: same collectives on every rank
    zero()
    backward(params[1:])
    red.finish()
    ok = ok and bool((params[0].grad == 0).all()) and torch.allclose(params[1].grad, mean_scale * params[1].detach(),
                                                                     atol=1e-6)
    # (5) close(): hooks gone, stream overlap back to what it was, idempotent
    red.close()
    red.close()
    zero()
    backward()
    ok = ok and all(w is None for w in red.work) and streams.ENABLED == was and not red.active
    ok = ok and all(torch.allclose(p.grad, 2.0 * (rank + 1) * p.detach(), atol=1e-6) for p in params)
    out[rank] = ok
    dist.destroy_process_group()


def test_reducer_semantics_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_semantics_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def _unused_worker(rank, world, port, out):
    """Only rank 0 leaves a parameter of bucket 0 (decoder) without a gradient: rank 1 launches every bucket
    from its hooks during backward, rank 0 cannot launch bucket 0 before finish().  With a launch order that
    followed completion, rank 0 would issue buckets 1, 2, 0 against rank 1's 0, 1, 2 (sizes differ: gloo
    raises or mixes buckets, RCCL hangs); with the global bucket order both issue 0, 1, 2."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=8, num_encoder_layer=2, lmfb_dim=4, num_classes=7, frame_stacking=1)
    torch.manual_seed(7)
    m = AttModel(hp)
    params = list(m.parameters())
    red = BucketedGradReducer(m)
    red.broadcast_parameters()
    unused = red.buckets[0][0]                    # a decoder parameter
    launched = []
    orig = red._launch
    red._launch = lambda bi: (launched.append(bi), orig(bi))[1]
    for p in params:
        p.grad = None
    used = [p for p in params if not (rank == 0 and p is unused)]
    ((rank + 1.0) * sum((p * p).sum() for p in used)).backward()
    during_backward = list(launched)
    red.finish()
    ok = launched == list(range(len(red.buckets)))                       # bucket order, on both ranks
    ok = ok and (during_backward == [] if rank == 0 else during_backward == launched)
    for p in params:
        # mean over ranks of d/dp (r+1) p^2 = 2 (r+1) p; rank 0 contributes zeros for `unused`
        scale = (2.0 * 2) / world if p is unused else sum(2.0 * (r + 1) for r in range(world)) / world
        ok = ok and bool(torch.allclose(p.grad, scale * p.detach(), atol=1e-6))
    red.close()
    out[rank] = ok
    dist.destroy_process_group()


def test_unused_parameter_on_one_rank_only():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_unused_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)
