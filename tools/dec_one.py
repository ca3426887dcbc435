"""Runs the decoder forward+backward once at config-2 shape (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_hp, make_batch
from asr_amd.Models.decoder import Decoder
dev = torch.device("cuda:0")
hp = make_hp(encoder_dropout=0.0)
torch.manual_seed(0)
dec = Decoder(hp).to(dev)
x, lengths, tgt, tl = make_batch(hp, dev)
L = int(os.environ.get("DEC_L", "6"))
h = torch.randn(32, 500, 640, device=dev, requires_grad=True)
y = dec(h, lengths, tgt[:, :L].contiguous())
y.sum().backward()
torch.cuda.synchronize()
