"""Drop-in for Models/AttModel.py:10-39 (BLSTM-encoder branch) on the HIP kernels.

    model = AttModel(hp)
    youtput = model(x, lengths, targets)        # (B,L,V), differentiable
    tokens  = model.decode(x, lengths)          # greedy / beam search, batch 1
    tokens  = model.decode_v2(x, lengths)       # the second beam search, batch 1

``forward(x, lengths, targets, src_pos=None)`` keeps the reference's argument
meaning: x (B,T,D) fp32 on the GPU (already frame-stacked), lengths (B,) int64
(on the CPU, any order), targets (B,L) int64 on the GPU.  The state_dict holds the
same 31 tensors with the same names, so ``network.epochN`` checkpoints load
either way.
"""
import torch
import torch.nn as nn

from ..utils.utils import lengths_to_device
from .decoder import Decoder
from .encoder import Encoder


class AttModel(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        enc = getattr(hp, "encoder_type", None)
        if enc is not None and enc != "BLSTM":
            raise NotImplementedError(
                f"asr_amd.AttModel: encoder_type={enc!r} is outside the accelerated path "
                "(SURVEY.md section 8: BLSTM encoder only)")
        # construction order = the reference's (encoder, then decoder) so a seeded
        # default init draws the same random numbers
        self.encoder = Encoder(hp)
        self.decoder = Decoder(hp)

    def forward(self, x, lengths, targets, src_pos=None):
        # one asynchronous host -> device hand-over of the lengths for both halves of the model
        lengths_dev = lengths_to_device(lengths, x.device)
        hbatch = self.encoder(x, lengths_dev, lengths_host=None if lengths.is_cuda else lengths)
        return self.decoder(hbatch, lengths_dev, targets)

    def decode(self, x, lengths, model_lm=None, src_pos=None):
        if model_lm is not None:
            raise NotImplementedError("asr_amd.AttModel.decode: LM shallow fusion is outside the path")
        with torch.no_grad():
            hbatch = self.encoder(x, lengths)
            return self.decoder.decode(hbatch, lengths, None)

    def decode_v2(self, x, lengths, model_lm=None, src_pos=None):
        """Models/AttModel.py:41-49."""
        if model_lm is not None:
            raise NotImplementedError("asr_amd.AttModel.decode_v2: LM shallow fusion is outside the path")
        with torch.no_grad():
            hbatch = self.encoder(x, lengths)
            return self.decoder.decode_v2(hbatch, lengths, None)
