"""Parameter holder for the location-aware attention -- Models/attention.py:6-22.

The arithmetic of Models/attention.py:24-40 runs inside the fused decoder
kernels (asr_amd/csrc/attdec.hip); this module only owns the five tensors, with
the reference's names, shapes and default initialisation, so checkpoints
interchange.
"""
import torch.nn as nn


class Attention(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.num_decoder_hidden_nodes = hp.num_hidden_nodes_decoder
        self.num_encoder_hidden_nodes = hp.num_hidden_nodes_encoder
        hd = hp.num_hidden_nodes_decoder
        he = hp.num_hidden_nodes_encoder * (1 if getattr(hp, "encoder_type", None) == "Conformer" else 2)   # attention.py:13-16, decoder.py:19-22
        # same construction order as the reference so a seeded init matches it
        self.L_se = nn.Linear(hd, hd * 2, bias=False)
        self.L_he = nn.Linear(he, hd * 2)
        self.L_ee = nn.Linear(hd * 2, 1, bias=False)
        self.L_fe = nn.Linear(10, hd * 2, bias=False)
        self.F_conv1d = nn.Conv1d(1, 10, 100, stride=1, padding=50, bias=False)

    def forward(self, *args, **kwargs):
        raise RuntimeError("asr_amd.Attention has no standalone forward: it runs inside the fused decoder "
                           "kernels (Decoder.forward / Decoder.decode)")
