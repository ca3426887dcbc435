"""Drop-in for Models/AttModel.py:10-49 (BLSTM and Conformer encoder branches) on the HIP kernels.

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
from .Transformer_encoder import TransformerEncoder


class AttModel(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        enc = getattr(hp, "encoder_type", None)
        if enc == "Wave":
            raise NotImplementedError(
                "asr_amd.AttModel: encoder_type='Wave' is outside the accelerated path "
                "(SURVEY.md section 8: BLSTM and Conformer encoders)")
        # construction order = the reference's (encoder, then decoder) so a seeded
        # default init draws the same random numbers
        if enc == "Conformer":
            self.encoder = TransformerEncoder(hp.lmfb_dim, 256, 16, 4, 0.1)     # Models/AttModel.py:17
        else:
            self.encoder = Encoder(hp)
        self.decoder = Decoder(hp)

    def _encode(self, x, lengths, src_pos):
        if getattr(self.hp, "encoder_type", None) == "Conformer":
            # Models/AttModel.py:23-27: the decoder receives the *un-subsampled* lengths, so its frame
            # mask is all ones wherever they exceed the subsampled length (the kernels clamp to T)
            return self.encoder(x, src_pos)
        return None

    def forward(self, x, lengths, targets, src_pos=None):
        # one asynchronous host -> device hand-over of the lengths for both halves of the model
        lengths_dev = lengths_to_device(lengths, x.device)
        hbatch = self._encode(x, lengths, src_pos)
        if hbatch is None:
            hbatch = self.encoder(x, lengths_dev, lengths_host=None if lengths.is_cuda else lengths)
        return self.decoder(hbatch, lengths_dev, targets)

    def decode(self, x, lengths, model_lm=None, src_pos=None):
        """Models/AttModel.py:31-39; ``model_lm``: shallow fusion with ``hp.lm_weight`` (Decoder.decode)."""
        with torch.no_grad():
            hbatch = self._encode(x, lengths, src_pos)
            if hbatch is None:
                hbatch = self.encoder(x, lengths)
            return self.decoder.decode(hbatch, lengths, model_lm)

    def decode_v2(self, x, lengths, model_lm=None, src_pos=None):
        """Models/AttModel.py:41-49."""
        with torch.no_grad():
            hbatch = self._encode(x, lengths, src_pos)
            if hbatch is None:
                hbatch = self.encoder(x, lengths)
            return self.decoder.decode_v2(hbatch, lengths, model_lm)
