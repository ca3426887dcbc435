"""GPU: the bucketed reducer on the Conformer branch in an RCCL group of one.  (Round 1 kept this file at the end
of the collection because side-stream tests stalled behind it; the cause -- a high-priority helper stream waiting on
the normal-priority main stream, asr_amd/parallel.py -- is gone, and it runs in normal collection order, with
other GPU tests behind it.)"""
import pytest
import torch

from tests._util import rel_err

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def test_conformer_data_parallel_rehearsal_group_of_one():
    """The N-GPU code path of asr_amd.parallel on the Conformer branch in an RCCL group of one (as bench.py's
    ASR_BENCH_FORCE_DP rehearsal does for config 2): hooks, bucket packing from the side stream, all-reduce, mean,
    scatter back -- the gradients must equal those of a plain backward."""
    import os
    import socket
    import torch.distributed as dist
    from asr_amd import streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=20, frame_stacking=1, num_hidden_nodes_encoder=256, num_hidden_nodes_decoder=24,
                   num_classes=31, encoder_type="Conformer", encoder_dropout=0.0)
    torch.manual_seed(4)
    m = AttModel(hp)
    for mod in m.modules():
        if hasattr(mod, "p"):
            mod.p = 0.0
    m = m.to(dev).train()
    B, T, L = 2, 90, 4
    x = torch.randn(B, T, 20, device=dev)
    src_pos = torch.arange(1, T + 1, device=dev).repeat(B, 1)
    targets = torch.randint(0, 31, (B, L), device=dev)
    lengths, tl = torch.tensor([T, T]), torch.tensor([4, 3], device=dev)

    def grads():
        m.zero_grad(set_to_none=True)
        # (training mode: BatchNorm uses batch statistics, so the running buffers do not enter the gradients)
        label_smoothing_loss(m(x, lengths, targets, src_pos), targets, tl).backward()

    grads()
    torch.cuda.synchronize()
    want = {k: p.grad.clone() for k, p in m.named_parameters()}
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    was = streams.ENABLED
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        with BucketedGradReducer(m, force=True) as red:
            assert red.active and len(red.buckets) == 1 + 4 + 1
            red.broadcast_parameters()
            grads()
            red.finish()
            torch.cuda.synchronize()
            assert all(w is None for w in red.work)
            for k, p in m.named_parameters():
                assert rel_err(p.grad.cpu(), want[k].cpu()) < 1e-6, k
            assert red.bytes_per_step() == 4 * sum(p.numel() for p in m.parameters())
        assert streams.ENABLED == was
    finally:
        dist.destroy_process_group()
